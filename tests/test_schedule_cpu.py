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
