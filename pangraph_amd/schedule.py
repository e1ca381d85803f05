"""Ready-set scheduling of a whole build over the native batch entry (include/pga_align.h).

The reference walks the guide tree in post-order, one merge after the other (packages/pangraph/src/commands/build/build_run.rs:111-128);
every merge runs the self-merge loop, i.e. `find_matches` on the joined child graphs (round 0), again on the merged graph (round 1), until
nothing matches (packages/pangraph/src/pangraph/graph_merging.rs:26-69,95-128).  The only true dependencies are

    (v, round 0)  needs the final round of both children of v        (leaves need nothing)
    (v, round r)  needs (v, round r - 1)

A level-synchronous host (the WAVES of pangraph_amd/levels.py) adds 42 barriers to that: every wave waits for its slowest whole-genome
query while most of the device idles.  Here a `find_matches` call (a TASK: one group of the batch entry) becomes ready the moment its
dependencies are done; up to `slots` batches are in flight at a time, each made of the ready tasks of that moment (largest remaining
path to the root first, at most `cap` bases per batch so that a large ready set is spread over the slots instead of queueing behind
one call).  The batches are ordinary pga_batch_create / pga_batch_align calls from different host threads: the library gives every call
its own stream and device-memory arena.  Results do not depend on how tasks are batched (groups are independent problems).

Multi-GPU (`world` > 1): the tree is cut into subtrees (`partition_subtrees`), every rank builds its subtrees with its own ready-set
scheduler and no communication, the match lists are gathered once; the merges above the cut (owner -1) are few, large and hang one
below the other: ALL ranks run them together, level by level -- every rank indexes the whole call and maps a contiguous range of the
queries of every group (pga_batch_align_shard) -- and a second gather ends the step (bench.py:step_ready).  `predict_scaling` turns the
partition and a cost model calibrated on one GPU into the step time to expect at N ranks.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


@dataclass
class Task:
    tid: int
    node: int                  # guide-tree node whose merge this call belongs to
    round: int                 # self-merge round
    seqs: list                 # 1-D uint8 arrays (views)
    names: List[str]
    deps: List[int]
    prio: float = 0.0          # estimated cost of the dependency chain from this task to the root (larger = earlier)
    bases: int = 0
    users: List[int] = field(default_factory=list)
    # flat C views, filled by prepare()
    ptr: Optional[np.ndarray] = None
    lens: Optional[np.ndarray] = None
    nptr: Optional[np.ndarray] = None
    _keep: Optional[list] = None

    def prepare(self):
        if self.ptr is not None:
            return
        for a in self.seqs:
            if a.dtype.itemsize != 1 or a.ndim != 1 or (len(a) and a.strides[0] != 1):
                raise ValueError("sequence arrays must be contiguous 1-D uint8")
        nb = [n.encode() for n in self.names]
        self._keep = nb
        self.ptr = np.fromiter((a.ctypes.data for a in self.seqs), dtype=np.uint64, count=len(self.seqs))
        self.lens = np.fromiter((len(a) for a in self.seqs), dtype=np.uint32, count=len(self.seqs))
        self.nptr = np.fromiter((C.cast(C.c_char_p(b), C.c_void_p).value for b in nb), dtype=np.uint64, count=len(nb))


def cost_estimate(bases: int, n_seqs: int) -> float:
    """seconds a task takes alone on the device (rough: a whole-genome pair is bound by its dependency chains, a block set by throughput;
    round 4, dev/path_probe.py, every call alone on the device: a spine call of two block sets 12-48 ms, its second round 6-10 ms, a
    whole-genome pair 63 ms; the longest chain of the BASELINE build measures 0.71 s, 0.51 s of it above tree height 4)"""
    big = bases / max(1, n_seqs)
    return 0.006 + bases * 2.5e-10 + (0.05 if big > 1e6 else 0.0)


def build_tasks(pop, min_block: int = 100, rounds: int = 2) -> List[Task]:
    """the find_matches calls of a simulated build (pangraph_amd.levels.Population) with their dependencies"""
    blocks = pop.clade_blocks(min_block)
    tasks: List[Task] = []
    last: Dict[int, int] = {}            # node -> tid of its final round
    for nd in reversed(pop.nodes):       # children have larger ids than their parent
        if not nd.children:
            continue
        c1, c2 = nd.children
        deps = [last[c] for c in (c1, c2) if c in last]
        t0 = Task(len(tasks), nd.id, 0, blocks[c1][0] + blocks[c2][0], blocks[c1][1] + blocks[c2][1], deps)
        tasks.append(t0)
        prev = t0
        for r in range(1, rounds):
            t = Task(len(tasks), nd.id, r, blocks[nd.id][0], blocks[nd.id][1], [prev.tid])
            tasks.append(t)
            prev = t
        last[nd.id] = prev.tid
    finish(tasks)
    return tasks


def finish(tasks: List[Task]) -> None:
    """users, base counts and priorities (longest remaining path, by estimated cost)"""
    for t in tasks:
        t.users = []
        t.bases = int(sum(len(a) for a in t.seqs))
    for t in tasks:
        for d in t.deps:
            tasks[d].users.append(t.tid)
    order = topo_order(tasks)
    for tid in reversed(order):
        t = tasks[tid]
        t.prio = cost_estimate(t.bases, len(t.seqs)) + max((tasks[u].prio for u in t.users), default=0.0)


def topo_order(tasks: List[Task]) -> List[int]:
    indeg = [len(t.deps) for t in tasks]
    ready = [t.tid for t in tasks if not t.deps]
    out = []
    while ready:
        x = ready.pop()
        out.append(x)
        for u in tasks[x].users:
            indeg[u] -= 1
            if indeg[u] == 0:
                ready.append(u)
    if len(out) != len(tasks):
        raise ValueError("dependency cycle")
    return out


class TaskBatch:
    """The flat C arrays pga_batch_create wants, for a list of tasks (one group per task); no copy of the bases."""

    def __init__(self, tasks: Sequence[Task], lib_first: Optional[Sequence[int]] = None):
        """lib_first: the sequences of task t are sequences lib_first[t.tid], lib_first[t.tid] + 1, ... of a batch that is already resident
        (batch.ResidentBatch(tb, derive_from=that batch)): no base pointer is handed over, `src` holds the indices (pga_batch_derive)."""
        for t in tasks:
            t.prepare()
        self.tasks = list(tasks)
        self.n_groups = len(tasks)
        self.src = None
        if lib_first is not None:
            self._src = np.concatenate([np.arange(lib_first[t.tid], lib_first[t.tid] + len(t.seqs), dtype=np.int64) for t in tasks]) if tasks else np.zeros(0, np.int64)
            self.src = self._src.ctypes.data_as(C.POINTER(C.c_int64))
        self._ptr = np.concatenate([t.ptr for t in tasks]) if tasks else np.zeros(0, np.uint64)
        if lib_first is not None:
            self._ptr = np.zeros_like(self._ptr)
        self._lens = np.concatenate([t.lens for t in tasks]) if tasks else np.zeros(0, np.uint32)
        self._nptr = np.concatenate([t.nptr for t in tasks]) if tasks else np.zeros(0, np.uint64)
        self._off = np.zeros(self.n_groups + 1, dtype=np.int64)
        self._off[1:] = np.cumsum([len(t.seqs) for t in tasks])
        self.seqs = self._ptr.ctypes.data_as(C.POINTER(C.c_char_p))
        self.cnames = self._nptr.ctypes.data_as(C.POINTER(C.c_char_p))
        self.lens = self._lens.ctypes.data_as(C.POINTER(C.c_uint32))
        self.off = self._off.ctypes.data_as(C.POINTER(C.c_int64))
        self.total_bases = int(sum(t.bases for t in tasks))

    @property
    def names(self):
        return [n for t in self.tasks for n in t.names]


class ReadySet:
    """The DECISIONS of the ready-set schedule without threads or clocks: which calls are ready, what may start now, which calls a batch takes,
    what a finished batch releases.  `run_ready_set` drives it from host threads; the library holds the same logic behind a C-ABI for hosts
    that are not Python (include/pga_sched.h, pangraph_amd/csrc/pga_sched.cpp: pga_sched_take / pga_sched_finish), and tests/test_schedule_cpu.py
    steps both through the same simulated builds and compares every batch."""

    def __init__(self, tasks: List[Task], slots: int = 6, cap_bases: float = 1.2e9, min_batch_bases: float = 0.0, done: Optional[set] = None,
                 only: Optional[set] = None, express: int = 0, express_eps: float = 0.05, express_cap: float = 60e6):
        self.tasks = tasks
        self.slots, self.cap_bases, self.min_batch_bases = slots, cap_bases, min_batch_bases
        self.express_eps, self.express_cap = express_eps, express_cap
        if slots < 1:
            raise ValueError("slots must be at least 1")
        self.fin = set(done or ())
        want = (set(range(len(tasks))) if only is None else set(only)) - self.fin      # a task given as done is not handed out again
        self.indeg = {}
        for tid in sorted(want):
            self.indeg[tid] = sum(1 for d in tasks[tid].deps if d not in self.fin)
            for d in tasks[tid].deps:
                if d not in self.fin and d not in want:
                    raise ValueError(f"task {tid} depends on {d}, which is neither done nor scheduled")
        self.ready = [tid for tid in sorted(want) if self.indeg[tid] == 0]
        self.left = len(want)
        self.in_flight = 0
        self.express = max(0, min(int(express), max(0, slots - 1)))
        self.n_express = 0                              # express batches in flight
        self.unfinished = sorted(want - self.fin, key=lambda tid: -tasks[tid].prio)   # for the longest remaining path
        self.pos_top = 0

    def crit_level(self) -> float:
        while self.pos_top < len(self.unfinished) and self.unfinished[self.pos_top] in self.fin:
            self.pos_top += 1
        return self.tasks[self.unfinished[self.pos_top]].prio if self.pos_top < len(self.unfinished) else 0.0

    def can_take(self) -> Optional[str]:
        """what may start now: "express" (a critical call is ready and an express slot is free), "bulk" (a bulk slot is free), or None"""
        tasks = self.tasks
        if not self.ready:
            return None
        if self.express and self.n_express < self.express and max(tasks[tid].prio for tid in self.ready) >= self.crit_level() - self.express_eps:
            return "express"
        if self.in_flight - self.n_express < self.slots - self.express:
            return "bulk"
        return None

    def take(self, kind: str) -> List[int]:
        tasks, ready = self.tasks, self.ready
        # largest remaining path first; stop at the cap (one oversized task still goes alone)
        ready.sort(key=lambda tid: -tasks[tid].prio)
        if kind == "express":
            lvl = self.crit_level() - self.express_eps
            got, b = [], 0
            for tid in ready:
                if tasks[tid].prio < lvl or (got and b + tasks[tid].bases > self.express_cap):
                    break
                got.append(tid); b += tasks[tid].bases
            taken = set(got)
            ready[:] = [tid for tid in ready if tid not in taken]
            self.in_flight += 1; self.n_express += 1
            return got
        total = sum(tasks[tid].bases for tid in ready)
        free = max(1, (self.slots - self.express) - (self.in_flight - self.n_express))
        cap = max(min(self.cap_bases, total / free if free > 1 else self.cap_bases), self.min_batch_bases, 1.0)
        got, b = [], 0
        rest = []
        for tid in ready:
            if not got or b + tasks[tid].bases <= cap:
                got.append(tid); b += tasks[tid].bases
            else:
                rest.append(tid)
        ready[:] = rest
        self.in_flight += 1
        return got

    def abandon(self, kind: str) -> None:
        """a batch that failed: its slot comes back, its calls stay unfinished"""
        self.in_flight -= 1
        self.n_express -= kind == "express"

    def finish(self, ids: Sequence[int], kind: str) -> None:
        self.in_flight -= 1
        self.n_express -= kind == "express"
        for i in ids:
            self.fin.add(i)
            self.left -= 1
            for u in self.tasks[i].users:
                if u in self.indeg:
                    self.indeg[u] -= 1
                    if self.indeg[u] == 0:
                        self.ready.append(u)


def run_ready_set(tasks: List[Task], run_batch: Callable[[List[Task]], object], slots: int = 6, cap_bases: float = 1.2e9,
                  min_batch_bases: float = 0.0, done: Optional[set] = None, only: Optional[set] = None,
                  on_result: Optional[Callable[[List[Task], object, float, float], None]] = None, express: int = 0, express_eps: float = 0.05,
                  express_cap: float = 60e6, native: Optional[bool] = None):
    """Runs `tasks` (all of them, or the subset `only`) in dependency order; `run_batch(list of tasks)` is called from up to `slots` host
    threads.  `done`: tids that count as finished from the start (results that arrived from elsewhere).  Returns the batch log
    [(t_start, t_end, n_tasks, bases)] relative to the start.

    express > 0: that many of the slots are an EXPRESS LANE for the critical path.  A call whose remaining path (Task.prio) is within
    `express_eps` seconds of the longest remaining path of the whole run is critical; it goes out the moment it is ready, alone or with the
    few other calls that are just as critical (at most `express_cap` bases), instead of waiting for a slot to come free and then sharing a batch
    -- and its latency -- with hundreds of Mbp of bulk work.  The bulk uses the other slots as before.

    native: the decisions come from the library's scheduler (pga_sched_*, the one a Rust or C++ host binds) instead of `ReadySet`; the
    threads block inside pga_sched_take.  Default: PGA_NATIVE_SCHED=1 in the environment."""
    if native is None:
        native = os.environ.get("PGA_NATIVE_SCHED", "0") not in ("", "0")
    if native:
        from . import sched_native
        return sched_native.run_ready_set(tasks, run_batch, slots=slots, cap_bases=cap_bases, min_batch_bases=min_batch_bases, done=done, only=only,
                                          on_result=on_result, express=express, express_eps=express_eps, express_cap=express_cap)
    rs = ReadySet(tasks, slots, cap_bases, min_batch_bases, done, only, express, express_eps, express_cap)
    cv = threading.Condition()
    log, errs = [], []
    t_origin = time.perf_counter()

    def worker():
        while True:
            with cv:
                kind = rs.can_take()
                while kind is None and rs.left > 0 and not errs:
                    cv.wait()
                    kind = rs.can_take()
                if rs.left <= 0 or errs:
                    return
                ids = rs.take(kind)
            t0 = time.perf_counter()
            try:
                res = run_batch([tasks[i] for i in ids])
                t1 = time.perf_counter()
                if on_result is not None:
                    on_result([tasks[i] for i in ids], res, t0 - t_origin, t1 - t_origin)
            except BaseException as e:   # noqa: BLE001
                with cv:
                    errs.append(e)
                    rs.abandon(kind)
                    cv.notify_all()
                return
            with cv:
                log.append((t0 - t_origin, t1 - t_origin, len(ids), sum(tasks[i].bases for i in ids)))
                rs.finish(ids, kind)
                cv.notify_all()

    th = [threading.Thread(target=worker, daemon=True) for _ in range(max(1, slots))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    return sorted(log)


# ---- the compiled host (pangraph_amd/host/build_driver.cpp): a build as a task file, its records back ------------------------------------
def write_task_file(tasks: List[Task], path: str, sensitivity: int = 10, n_threads: int = 8, pop=None) -> None:
    """the calls of a build -- dependencies, block names, block sequences, and (pop given) the guide tree a host of several ranks cuts into subtrees -- in
    the little-endian layout build_driver.cpp reads"""
    import struct
    with open(path, "wb") as f:
        n_nodes = len(pop.nodes) if pop is not None else 0
        f.write(b"PGAB1\0\0\0" + struct.pack("<4i", len(tasks), sensitivity, n_threads, n_nodes))
        if n_nodes:
            f.write(np.asarray([nd.children[0] if nd.children else -1 for nd in pop.nodes], dtype="<i4").tobytes())
            f.write(np.asarray([nd.children[1] if nd.children else -1 for nd in pop.nodes], dtype="<i4").tobytes())
            f.write(np.asarray([t.node for t in tasks], dtype="<i4").tobytes())
        for t in tasks:
            f.write(struct.pack("<i", len(t.deps)) + struct.pack(f"<{len(t.deps)}i", *t.deps) + struct.pack("<i", len(t.seqs)))
            for a, name in zip(t.seqs, t.names):
                nb = name.encode()
                f.write(struct.pack("<II", len(a), len(nb)) + nb)
                f.write(np.ascontiguousarray(a).tobytes())


def read_driver_results(path: str):
    """(per task: (pga_match_t records, CIGAR pool as uint32), number of batches) from build_driver's output file"""
    import struct
    from .dist import MATCH_DTYPE
    buf = open(path, "rb").read()
    if buf[:8] != b"PGAR1\0\0\0":
        raise ValueError(f"{path} is not a result file of build_driver")
    n, n_batches = struct.unpack_from("<2i", buf, 8)
    pos, out = 16, []
    for _ in range(n):
        (nm,) = struct.unpack_from("<q", buf, pos); pos += 8
        m = np.frombuffer(buf, dtype=MATCH_DTYPE, count=nm, offset=pos).copy(); pos += nm * MATCH_DTYPE.itemsize
        (nc,) = struct.unpack_from("<q", buf, pos); pos += 8
        c = np.frombuffer(buf, dtype=np.uint32, count=nc, offset=pos).copy(); pos += 4 * nc
        out.append((m, c))
    if pos != len(buf):
        raise ValueError(f"{path}: {len(buf) - pos} bytes behind the last task")
    return out, n_batches


# ---- multi-GPU: subtrees -------------------------------------------------------------------------------------------------------------
def partition_subtrees(pop, tasks: List[Task], world: int, per_rank: int = 4):
    """Cuts the guide tree into at least world * per_rank subtrees (splitting the heaviest one at its root until there are enough), deals
    them to the ranks heaviest first (LPT on base counts) and returns (owner per task: rank, or -1 for the merges above the cut that run on
    rank 0 after the gather; list of subtree roots per rank).  Deterministic: every rank computes the same plan without talking."""
    if world <= 1:
        return [0] * len(tasks), [[0]]
    node_tasks: Dict[int, List[int]] = {}
    for t in tasks:
        node_tasks.setdefault(t.node, []).append(t.tid)
    weight: Dict[int, float] = {}
    for nd in reversed(pop.nodes):
        w = sum(tasks[i].bases for i in node_tasks.get(nd.id, []))
        for c in nd.children:
            w += weight[c]
        weight[nd.id] = w
    roots = [0]
    top = set()                                   # internal nodes above the cut
    while len(roots) < world * per_rank:
        cand = [r for r in roots if pop.nodes[r].children]
        if not cand:
            break
        r = max(cand, key=lambda x: (weight[x], -x))
        roots.remove(r)
        top.add(r)
        roots += list(pop.nodes[r].children)
    roots = [r for r in roots if pop.nodes[r].children]     # a bare leaf holds no merge
    order = sorted(roots, key=lambda r: (-weight[r], r))
    load = [0.0] * world
    per = [[] for _ in range(world)]
    for r in order:
        k = min(range(world), key=lambda i: (load[i], i))
        per[k].append(r)
        load[k] += weight[r]
    owner_of_node: Dict[int, int] = {}
    for k, rs in enumerate(per):
        for r in rs:
            stack = [r]
            while stack:
                x = stack.pop()
                owner_of_node[x] = k
                stack += list(pop.nodes[x].children)
    owner = [(-1 if t.node in top else owner_of_node[t.node]) for t in tasks]
    return owner, per


def predict_scaling(pop, tasks: List[Task], worlds: Sequence[int], gbp_s_one_gpu: float, slots: int = 6, host_cpu_s_per_gbp: float = 0.0,
                    host_cores: int = 0) -> Dict[str, dict]:
    """What the subtree partition lets N ranks do, from ONE GPU's measurements: a rank cannot finish phase 1 before (a) its share of the bases
    has gone through at the single-GPU throughput, (b) the longest dependency chain of its subtrees has run (cost_estimate per call, the
    calls of a chain one after the other) and (c) the HOST has done its part: the driver of a batch costs host_cpu_s_per_gbp core-seconds per Gbp
    (measured at N = 1: process CPU time of the timed steps / bases), and the N ranks of a node share its host_cores -- a rank gets host_cores / N of them
    (16 usable cores on the pool's boxes: two per rank at N = 8, where one rank alone keeps seven busy); phase 2 is the merges above the cut level by
    level, each level shortened by the query split only as far as the per-call floor allows, every rank paying the host cost of the WHOLE level's
    index on its share of the cores.  A MODEL to hold the first real multi-GPU run against -- not a measurement."""
    out = {}
    total = float(sum(t.bases for t in tasks))
    for n in worlds:
        owner, _ = partition_subtrees(pop, tasks, n)
        load = [0.0] * max(1, n)
        order = topo_order(tasks)
        longest = [0.0] * len(tasks)
        for tid in order:
            t = tasks[tid]
            if owner[tid] < 0:
                continue
            longest[tid] = cost_estimate(t.bases, len(t.seqs)) + max((longest[d] for d in t.deps if owner[d] == owner[tid]), default=0.0)
            load[owner[tid]] += t.bases
        crit = [max((longest[t.tid] for t in tasks if owner[t.tid] == r), default=0.0) for r in range(max(1, n))]
        cores_per_rank = (host_cores / max(1, n)) if host_cores else 0.0
        host1 = [(load[r] / 1e9 * host_cpu_s_per_gbp / cores_per_rank) if cores_per_rank else 0.0 for r in range(max(1, n))]
        phase1 = max(max(load[r] / 1e9 / gbp_s_one_gpu, crit[r], host1[r]) for r in range(max(1, n)))
        # phase 2 as bench.py runs it: the calls above the cut level by level (a level = the calls whose dependencies are done, ONE batch that
        # all ranks work on with the queries split): a level waits for its slowest call's floor, its bases go through n ranks
        top = [t for t in tasks if owner[t.tid] < 0]
        done = {t.tid for t in tasks if owner[t.tid] >= 0}
        left = [t.tid for t in top]
        phase2 = 0.0
        while left:
            level = [i for i in left if all(d in done for d in tasks[i].deps)]
            if not level:
                break
            floor = max(cost_estimate(0, 1) + (cost_estimate(tasks[i].bases, len(tasks[i].seqs)) - cost_estimate(0, 1)) / n for i in level)
            lvl_gbp = sum(tasks[i].bases for i in level) / 1e9
            host2 = (lvl_gbp * host_cpu_s_per_gbp / cores_per_rank) if cores_per_rank else 0.0     # (every rank indexes the whole level)
            phase2 += max(floor, lvl_gbp * 0.25 / n + cost_estimate(0, 1), host2)
            done.update(level)
            left = [i for i in left if i not in done]
        step = phase1 + phase2
        out[str(n)] = {"per_rank_gbp": [round(x / 1e9, 6) for x in load], "per_rank_critical_path_s": [round(x, 3) for x in crit], "calls_above_the_cut": len(top),
                       "phase1_s": round(phase1, 3), "phase2_s": round(phase2, 3), "step_s": round(step, 3), "gbp_s": round(total / 1e9 / step, 2) if step > 0 else None,
                       "host_cores_per_rank": round(cores_per_rank, 2) if cores_per_rank else None,
                       "phase1_bound": ("host" if host1 and max(host1) >= phase1 - 1e-12 and max(host1) > 0 else "critical path" if max(crit) >= phase1 - 1e-12 else "device throughput")}
    return out
