"""Host logic of the ready-set schedule (pangraph_amd/schedule.py): every find_matches call of a simulated build runs exactly once, never before
the calls it depends on (graph_merging.rs:26-69: a merge needs the graphs of both children; a self-merge round needs the previous one), the
batches respect the cap, and the subtree partition for N ranks covers every call once."""
import threading
import time

import numpy as np

from pangraph_amd import schedule as sched
from pangraph_amd.levels import Population, Rates, waves_bases


def _pop(n=24):
    return Population(11, n, 20_000, Rates(ev_min=200, ev_max=2000))


def test_tasks_are_the_waves_of_the_build():
    pop = _pop()
    tasks = sched.build_tasks(pop)
    waves = pop.build_waves()
    assert len(tasks) == sum(len(g) for _, g, _ in waves)
    assert sum(t.bases for t in tasks) == waves_bases(waves)
    # the same groups, wave by wave: (height, round) of a task names its wave
    by_wave = {}
    for t in tasks:
        by_wave.setdefault((pop.nodes[t.node].height, t.round), []).append(t)
    for w, (label, groups, names) in enumerate(waves):
        h, r = w // 2 + 1, w % 2
        got = sorted((tuple(t.names) for t in by_wave[(h, r)]))
        assert got == sorted(tuple(n) for n in names), label
    for t in tasks:
        for d in t.deps:
            assert tasks[d].prio > t.prio          # a dependency has the longer remaining path


def test_ready_set_respects_dependencies_and_cap():
    pop = _pop(40)
    tasks = sched.build_tasks(pop)
    done_at, started_at, lock = {}, {}, threading.Lock()
    sizes = []

    def run_batch(ts):
        with lock:
            now = time.perf_counter()
            for t in ts:
                assert t.tid not in started_at
                started_at[t.tid] = now
                for d in t.deps:
                    assert d in done_at, (t.tid, d)
            sizes.append(sum(t.bases for t in ts))
        tb = sched.TaskBatch(ts)
        assert tb.n_groups == len(ts) and tb.total_bases == sum(t.bases for t in ts)
        assert int(tb._off[-1]) == sum(len(t.seqs) for t in ts) == len(tb._ptr) == len(tb._lens) == len(tb._nptr)
        time.sleep(0.002)
        with lock:
            now = time.perf_counter()
            for t in ts:
                done_at[t.tid] = now
        return len(ts)

    cap = 150_000
    log = sched.run_ready_set(tasks, run_batch, slots=3, cap_bases=cap)
    assert sorted(done_at) == list(range(len(tasks)))
    assert sum(n for _, _, n, _ in log) == len(tasks)
    biggest = max(t.bases for t in tasks)
    assert all(s <= max(cap, biggest) for s in sizes)


def test_errors_surface():
    tasks = sched.build_tasks(_pop(8))

    def run_batch(ts):
        raise RuntimeError("boom")

    try:
        sched.run_ready_set(tasks, run_batch, slots=2)
    except RuntimeError as e:
        assert "boom" in str(e)
    else:
        raise AssertionError("no error")


def test_subtree_partition_covers_every_call_once():
    pop = _pop(60)
    tasks = sched.build_tasks(pop)
    for world in (2, 4, 8):
        owner, per = sched.partition_subtrees(pop, tasks, world)
        assert len(owner) == len(tasks) and set(owner) <= set(range(world)) | {-1}
        # a task of a rank depends only on tasks of the same rank; a task above the cut on anything
        for t in tasks:
            for d in t.deps:
                assert owner[t.tid] == -1 or owner[d] == owner[t.tid]
        assert sum(1 for o in owner if o == -1) <= 2 * (4 * world)          # only the merges above the cut
        load = [sum(t.bases for t in tasks if owner[t.tid] == r) for r in range(world)]
        assert max(load) <= 2.0 * (sum(load) / world) + max(t.bases for t in tasks)
        # phase 1 on every rank, then phase 2: everything runs once
        ran = []
        for r in range(world):
            sched.run_ready_set(tasks, lambda ts: ran.extend(t.tid for t in ts), slots=2, only={t.tid for t in tasks if owner[t.tid] == r})
        sched.run_ready_set(tasks, lambda ts: ran.extend(t.tid for t in ts), slots=2, only={t.tid for t in tasks if owner[t.tid] == -1},
                            done={t.tid for t in tasks if owner[t.tid] != -1})
        assert sorted(ran) == list(range(len(tasks)))


def test_more_ranks_than_subtrees_and_the_scaling_model():
    """8 ranks on a 10-genome tree: some ranks own nothing, most merges sit above the cut -- the plan must still cover every call once; and the
    scaling model (schedule.predict_scaling) reports, per N, loads that add up to the build and a step that is never shorter than its parts."""
    pop = _pop(10)
    tasks = sched.build_tasks(pop)
    owner, per = sched.partition_subtrees(pop, tasks, 8)
    assert len(per) == 8 and any(len(p) == 0 for p in per)               # empty ranks exist
    assert set(owner) <= set(range(8)) | {-1}
    ran = []
    for r in range(8):
        mine = {t.tid for t in tasks if owner[t.tid] == r}
        if mine:
            sched.run_ready_set(tasks, lambda ts: ran.extend(t.tid for t in ts), slots=2, only=mine)
    sched.run_ready_set(tasks, lambda ts: ran.extend(t.tid for t in ts), slots=2, only={t.tid for t in tasks if owner[t.tid] == -1},
                        done={t.tid for t in tasks if owner[t.tid] != -1})
    assert sorted(ran) == list(range(len(tasks)))
    big = _pop(60)
    bt = sched.build_tasks(big)
    model = sched.predict_scaling(big, bt, (1, 2, 4, 8), gbp_s_one_gpu=0.01)
    total = sum(t.bases for t in bt) / 1e9
    for n, m in model.items():
        above = sum(t.bases for t, o in zip(bt, sched.partition_subtrees(big, bt, int(n))[0]) if o == -1) / 1e9
        assert abs(sum(m["per_rank_gbp"]) + above - total) < 1e-2 * max(total, 1e-9) + 1e-3
        assert m["step_s"] >= m["phase1_s"] and m["step_s"] >= m["phase2_s"] and len(m["per_rank_gbp"]) == int(n)
    assert model["1"]["calls_above_the_cut"] == 0 and model["8"]["calls_above_the_cut"] > 0


def test_scaling_model_prices_the_host():
    """schedule.predict_scaling with the host term (core-seconds per Gbp measured at N = 1, the node's cores shared by its ranks): a step is never
    shorter than without it, at N = 1 a host that is not the bottleneck changes nothing, and with few cores per rank the model says so."""
    big = _pop(60)
    bt = sched.build_tasks(big)
    total = sum(t.bases for t in bt) / 1e9
    free = sched.predict_scaling(big, bt, (1, 2, 4, 8), gbp_s_one_gpu=0.01)
    # a host that needs a tenth of a core-second per device-second at N = 1: irrelevant there, the bound at N = 8 on two cores per rank
    cheap = sched.predict_scaling(big, bt, (1, 2, 4, 8), gbp_s_one_gpu=0.01, host_cpu_s_per_gbp=0.1 / 0.01 * 0.1, host_cores=16)
    dear = sched.predict_scaling(big, bt, (1, 2, 4, 8), gbp_s_one_gpu=0.01, host_cpu_s_per_gbp=7.0 / 0.01, host_cores=16)      # seven cores busy at N = 1
    for n in ("1", "2", "4", "8"):
        assert cheap[n]["step_s"] >= free[n]["step_s"] - 1e-9 and dear[n]["step_s"] >= cheap[n]["step_s"] - 1e-9
        assert dear[n]["host_cores_per_rank"] == 16 / int(n)
    assert abs(cheap["1"]["step_s"] - free["1"]["step_s"]) < 1e-9
    assert dear["8"]["phase1_bound"] == "host" and dear["8"]["gbp_s"] < 0.5 * free["8"]["gbp_s"]
    assert total > 0


# ---- the library's scheduler (include/pga_sched.h) against schedule.ReadySet ---------------------------------------------------------
def _simulate(tasks, taker, duration):
    """Steps a scheduler through a build without threads or clocks: whatever may start starts, then the batch that ends first ends.
    taker: object with try_take() -> (ids, handle) | ([], _) nothing may start | (None, _) run over, and finish(handle)."""
    now, flying, batches, seq = 0.0, [], [], 0
    while True:
        ids, h = taker.try_take()
        if ids:
            batches.append(list(ids))
            flying.append((now + duration(ids), seq, h)); seq += 1
            continue
        if not flying:
            assert ids is None or taker.left() == 0, "nothing in flight, nothing may start, yet calls are left"
            return batches
        flying.sort()
        now, _, h = flying.pop(0)
        taker.finish(h)


class _PyTaker:
    def __init__(self, tasks, **kw):
        self.rs = sched.ReadySet(tasks, **kw)

    def try_take(self):
        if self.rs.left <= 0:
            return None, None
        kind = self.rs.can_take()
        if kind is None:
            return [], None
        ids = self.rs.take(kind)
        return ids, (ids, kind)

    def finish(self, h):
        self.rs.finish(*h)

    def left(self):
        return self.rs.left


class _NativeTaker:
    def __init__(self, tasks, **kw):
        from pangraph_amd import sched_native
        self.ns = sched_native.NativeSched(tasks)
        self.ns.start(**kw)

    def try_take(self):
        return self.ns.try_take()

    def finish(self, h):
        self.ns.finish(h)

    def left(self):
        return self.ns.left()


def test_native_scheduler_cuts_the_same_batches(product_so):
    """pga_sched_* (what a Rust host binds) and schedule.ReadySet (what bench.py drives) take the same calls into the same batches in the same
    order, over whole simulated builds: slot counts, caps, the express lane, and the two phases of a multi-rank step (`only` / `done`)."""
    from pangraph_amd import sched_native
    pop = _pop(60)
    tasks = sched.build_tasks(pop)
    ns = sched_native.NativeSched(tasks)
    assert [float(p) for p in ns.prio()] == [t.prio for t in tasks]                      # bit for bit: same sums in the same order
    for b, n in ((0, 0), (1, 1), (123_456, 2), (5_000_001, 2), (10_000_000, 7), (4_000_000, 4)):
        assert sched_native.cost(b, n) == sched.cost_estimate(b, n)
    durs = {
        "cost": lambda ids: sum(sched.cost_estimate(tasks[i].bases, len(tasks[i].seqs)) for i in ids),
        "jitter": lambda ids: 0.001 + ((ids[0] * 2654435761) % 1000) * 1e-5,                # batches overtake each other
    }
    n_cmp = 0
    for name, dur in durs.items():
        for kw in (dict(slots=1), dict(slots=3, cap_bases=150_000), dict(slots=6, cap_bases=400_000, min_batch_bases=50_000),
                   dict(slots=6, cap_bases=1.2e9, express=1, express_eps=0.004, express_cap=90_000), dict(slots=4, cap_bases=300_000, express=2, express_eps=0.01),
                   dict(slots=2, cap_bases=1.0, express=5)):
            a = _simulate(tasks, _PyTaker(tasks, **kw), dur)
            b = _simulate(tasks, _NativeTaker(tasks, **kw), dur)
            assert a == b, (name, kw)
            assert sorted(i for ids in a for i in ids) == list(range(len(tasks)))
            n_cmp += len(a)
    # the phases of a multi-rank step
    for world in (2, 8):
        owner, _ = sched.partition_subtrees(pop, tasks, world)
        assert sched_native.partition(pop, tasks, world) == owner
        for r in range(world):
            mine = {t.tid for t in tasks if owner[t.tid] == r}
            if mine:
                kw = dict(slots=3, cap_bases=200_000, only=mine)
                assert _simulate(tasks, _PyTaker(tasks, **kw), durs["jitter"]) == _simulate(tasks, _NativeTaker(tasks, **kw), durs["jitter"])
        kw = dict(slots=3, cap_bases=200_000, only={t.tid for t in tasks if owner[t.tid] == -1}, done={t.tid for t in tasks if owner[t.tid] != -1})
        a = _simulate(tasks, _PyTaker(tasks, **kw), durs["cost"])
        assert a == _simulate(tasks, _NativeTaker(tasks, **kw), durs["cost"]) and sorted(i for ids in a for i in ids) == sorted(kw["only"])
    assert n_cmp > 200
    assert sched_native.partition(pop, tasks, 1) == [0] * len(tasks)
    for per_rank in (1, 2, 7):
        assert sched_native.partition(pop, tasks, 4, per_rank) == sched.partition_subtrees(pop, tasks, 4, per_rank)[0]


def test_native_scheduler_from_threads_and_its_errors(product_so):
    """the blocking entry from host threads (the way bench.py --native-sched and a Rust host use it): every call once, dependencies first; a
    failing batch ends the run; a dependency that is neither done nor scheduled, a cycle and a bad id are errors with a text."""
    import pytest
    from pangraph_amd import sched_native
    tasks = sched.build_tasks(_pop(40))
    done_at, lock = {}, threading.Lock()

    def run_batch(ts):
        with lock:
            for t in ts:
                assert t.tid not in done_at
                for d in t.deps:
                    assert d in done_at, (t.tid, d)
        time.sleep(0.001)
        with lock:
            for t in ts:
                done_at[t.tid] = time.perf_counter()
        return len(ts)

    seen = []
    log = sched.run_ready_set(tasks, run_batch, slots=4, cap_bases=150_000, native=True, on_result=lambda ts, res, t0, t1: seen.append(res))
    assert sorted(done_at) == list(range(len(tasks))) and sum(n for _, _, n, _ in log) == len(tasks) == sum(seen)

    def boom(ts):
        raise RuntimeError("boom")
    with pytest.raises(RuntimeError, match="boom"):
        sched.run_ready_set(tasks, boom, slots=3, native=True)
    ns = sched_native.NativeSched(tasks)
    with pytest.raises(ValueError, match="neither done nor scheduled"):
        ns.start(only={t.tid for t in tasks if t.deps})
    with pytest.raises(ValueError, match="out of range"):
        ns.start(only={len(tasks) + 5})

    class T:
        def __init__(self, deps):
            self.deps, self.bases, self.seqs = deps, 10, [b"A"]
    with pytest.raises(ValueError, match="cycle"):
        sched_native.NativeSched([T([1]), T([0])])
    with pytest.raises(ValueError, match="bad dependency"):
        sched_native.NativeSched([T([0])])


def test_native_scheduler_small_buffer_and_abort(product_so):
    """pga_sched_take with a buffer that cannot hold the batch takes nothing (-1, the size needed in *ticket) and the next call with room gets the very
    batch the first would have got; after pga_sched_abort every take returns 0."""
    import ctypes as C
    import numpy as np
    from pangraph_amd import sched_native
    tasks = sched.build_tasks(_pop(40))
    d = sched_native.lib()
    want = _simulate(tasks, _PyTaker(tasks, slots=2, cap_bases=1.2e9), lambda ids: 1.0)
    assert len(want[0]) > 4                                                    # the first batch: half of the leaf calls (two free slots)
    ns = sched_native.NativeSched(tasks)
    ns.start(slots=2, cap_bases=1.2e9)
    small = np.zeros(2, dtype=np.int32)
    ticket = C.c_int32(-7)
    assert d.pga_sched_try_take(ns.h, small.ctypes.data_as(sched_native.I32P), 2, C.byref(ticket)) == -1 and ticket.value == len(want[0])
    assert d.pga_sched_take(ns.h, small.ctypes.data_as(sched_native.I32P), 2, C.byref(ticket)) == -1 and ticket.value == len(want[0])
    assert ns.left() == len(tasks)
    ids, t = ns.try_take()
    assert ids == want[0]
    ids2, t2 = ns.try_take()                                                   # the second slot: the rest of the ready set
    assert ids2 == want[1] and t2 != t
    assert ns.try_take() == ([], -1)                                           # both slots taken
    ns.finish(t)
    ns.finish(t)                                                               # a ticket is good for one finish
    ns.finish(12345)
    assert ns.left() == len(tasks) - len(ids)
    ns.abort()
    assert ns.take() == (None, -1) and ns.try_take() == (None, -1)
    ns.start(only=set())                                                       # an empty run is over at once (not "all tasks")
    assert ns.left() == 0 and ns.take() == (None, -1)
    assert sched.run_ready_set(tasks, lambda ts: 1 / 0, slots=2, only=set(), native=True) == []
    ns.close()


def test_compiled_host_loop_without_a_device(product_so, tmp_path):
    """pangraph_amd/host/build_driver (a whole build driven from C++: task file, pga_sched_*, worker threads, result file) with PGA_DRIVER_DRY=1 -- no
    device work, every call "finds" nothing: all calls run, the files round-trip; without the switch and without a GPU it says so (exit 3)."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "pangraph_amd", "host", "build_driver")
    assert os.path.exists(exe), "built by make -C pangraph_amd/csrc"
    tasks = sched.build_tasks(_pop(24))
    tf, of = str(tmp_path / "tasks.bin"), str(tmp_path / "out.bin")
    sched.write_task_file(tasks, tf)
    assert os.path.getsize(tf) > sum(t.bases for t in tasks)
    r = subprocess.run([exe, tf, of, "4", "150000"], env=dict(os.environ, PGA_DRIVER_DRY="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    res, n_batches = sched.read_driver_results(of)
    assert len(res) == len(tasks) and all(len(m) == 0 and len(c) == 0 for m, c in res)
    want = _simulate(tasks, _PyTaker(tasks, slots=1, cap_bases=150_000), lambda ids: 1.0)
    assert n_batches >= len(want) // 4 and f"{len(tasks)} calls in {n_batches} batches" in r.stdout
    (tmp_path / "bad.bin").write_bytes(b"PGAB1\0\0\0" + b"\xff" * 7)
    assert subprocess.run([exe, str(tmp_path / "bad.bin"), of], capture_output=True, text=True).returncode == 2
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, tf, of], capture_output=True, text=True, timeout=120)
        assert r.returncode == 3 and "no HIP device" in r.stderr


def test_compiled_host_with_several_ranks_without_a_device(product_so, tmp_path):
    """build_driver as THREE ranks (PGA_RANK / PGA_WORLD / PGA_XDIR; three processes, PGA_DRIVER_DRY=1): every rank cuts the guide tree the task file
    carries with pga_sched_partition, runs the calls of its subtrees under its own schedule, publishes its list, walks the calls above the cut level by
    level and publishes again; rank 0 merges and writes the result file.  Every call is accounted for exactly once -- the per-rank counts are those of
    schedule.partition_subtrees -- and a task file without the tree, or a rank outside the world, is refused."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "pangraph_amd", "host", "build_driver")
    pop = _pop(24)
    tasks = sched.build_tasks(pop)
    tf, of, xd = str(tmp_path / "tasks.bin"), str(tmp_path / "out.bin"), tmp_path / "x"
    xd.mkdir()
    sched.write_task_file(tasks, tf, pop=pop)
    world = 3
    owner, _ = sched.partition_subtrees(pop, tasks, world)
    procs = [subprocess.Popen([exe, tf, of if r == 0 else str(tmp_path / f"unused{r}.bin"), "2", "150000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, PGA_DRIVER_DRY="1", PGA_RANK=str(r), PGA_WORLD=str(world), PGA_XDIR=str(xd), PGA_XTIMEOUT_S="60")) for r in range(world)]
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1] for o in outs]
    n_above = sum(1 for o in owner if o < 0)
    assert f"{world} ranks: {len(tasks)} calls ({n_above} above the cut)" in outs[0][0], outs[0][0]
    for r in range(1, world):
        assert f"rank {r} of {world}: {sum(1 for o in owner if o == r)} calls of its own" in outs[r][0], outs[r][0]
    res, _ = sched.read_driver_results(of)
    assert len(res) == len(tasks) and not os.path.exists(tmp_path / "unused1.bin")
    assert sorted(os.listdir(xd)) == sorted(f"phase{p}_{r}.bin" for p in (1, 2) for r in range(world))
    sched.write_task_file(tasks, str(tmp_path / "notree.bin"))
    env = dict(os.environ, PGA_DRIVER_DRY="1", PGA_RANK="0", PGA_WORLD="2", PGA_XDIR=str(xd))
    assert subprocess.run([exe, str(tmp_path / "notree.bin"), of], env=env, capture_output=True, text=True).returncode == 2
    assert subprocess.run([exe, tf, of], env=dict(env, PGA_RANK="2"), capture_output=True, text=True).returncode == 2
