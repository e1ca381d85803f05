"""ctypes mirror of include/pga_sched.h: the ready-set schedule held by the library (pangraph_amd/csrc/pga_sched.cpp), i.e. what a host that is
not Python binds.  `run_ready_set` here is `schedule.run_ready_set` with the decisions taken inside the library: the
worker threads block in pga_sched_take (outside the interpreter lock) instead of on a Python condition variable.

No device work happens behind these entry points; the library loads without a GPU."""
from __future__ import annotations

import ctypes as C
import threading
import time
from typing import Callable, List, Optional, Sequence

import numpy as np

from .batch import LIB_PATH, PgaError

_lib = None
I32P, I64P, F64P = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        import os
        if not os.path.exists(LIB_PATH):
            raise PgaError(f"{LIB_PATH} is missing: build it with __graft_entry__.build()")
        d = C.CDLL(LIB_PATH)
        d.pga_sched_create.restype = C.c_void_p
        d.pga_sched_create.argtypes = [C.c_int32, I64P, I32P, I64P, I32P]
        d.pga_sched_destroy.argtypes = [C.c_void_p]
        d.pga_sched_error.restype = C.c_char_p
        d.pga_sched_cost.restype = C.c_double
        d.pga_sched_cost.argtypes = [C.c_int64, C.c_int32]
        d.pga_sched_prio.argtypes = [C.c_void_p, F64P]
        d.pga_sched_start.restype = C.c_int
        d.pga_sched_start.argtypes = [C.c_void_p, I32P, C.c_int32, I32P, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_double, C.c_double]
        for f in (d.pga_sched_take, d.pga_sched_try_take):
            f.restype = C.c_int32
            f.argtypes = [C.c_void_p, I32P, C.c_int32, I32P]
        d.pga_sched_finish.argtypes = [C.c_void_p, C.c_int32]
        d.pga_sched_abort.argtypes = [C.c_void_p]
        d.pga_sched_left.restype = C.c_int32
        d.pga_sched_left.argtypes = [C.c_void_p]
        d.pga_sched_partition.restype = C.c_int32
        d.pga_sched_partition.argtypes = [C.c_int32, I32P, I32P, C.c_int32, I32P, I64P, C.c_int32, C.c_int32, I32P]
        _lib = d
    return _lib


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(I32P)


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(I64P)


class NativeSched:
    """One task graph in the library.  tasks: objects with .deps, .bases and .seqs (schedule.Task)."""

    def __init__(self, tasks: Sequence):
        self.n = len(tasks)
        off = np.zeros(self.n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(t.deps) for t in tasks])
        dep = [d for t in tasks for d in t.deps]
        self._k = [_i64(off), _i32(dep if dep else [0]), _i64([t.bases for t in tasks] or [0]), _i32([len(t.seqs) for t in tasks] or [0])]
        self.h = lib().pga_sched_create(self.n, self._k[0][1], self._k[1][1], self._k[2][1], self._k[3][1])
        if not self.h:
            raise ValueError(lib().pga_sched_error().decode())
        self._buf = threading.local()

    def close(self):
        if self.h:
            lib().pga_sched_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass

    def prio(self) -> np.ndarray:
        out = np.zeros(max(1, self.n), dtype=np.float64)
        lib().pga_sched_prio(self.h, out.ctypes.data_as(F64P))
        return out[:self.n]

    def start(self, slots: int = 6, cap_bases: float = 1.2e9, min_batch_bases: float = 0.0, done=None, only=None, express: int = 0,
              express_eps: float = 0.05, express_cap: float = 60e6) -> None:
        # (an empty `only` is an empty run, not "all tasks": the array handed over is never NULL then)
        n_only = 0 if only is None else len(only)
        o = _i32(sorted(only) or [0]) if only is not None else (None, None)
        dn = _i32(sorted(done)) if done else (None, None)
        rc = lib().pga_sched_start(self.h, o[1], n_only, dn[1], 0 if dn[0] is None else len(dn[0]), int(slots), float(cap_bases),
                                   float(min_batch_bases), int(express), float(express_eps), float(express_cap))
        if rc != 0:
            raise ValueError(lib().pga_sched_error().decode())

    def _take(self, fn):
        b = getattr(self._buf, "ids", None)
        if b is None:
            b = self._buf.ids = np.zeros(max(1, self.n), dtype=np.int32)
        ticket = C.c_int32(-1)
        n = fn(self.h, b.ctypes.data_as(I32P), len(b), C.byref(ticket))
        if n <= 0:
            return n, None, -1
        return n, [int(x) for x in b[:n]], ticket.value

    def take(self):
        """blocks; (ids, ticket), or (None, -1) when the run is over"""
        n, ids, ticket = self._take(lib().pga_sched_take)
        return (ids, ticket) if n > 0 else (None, -1)

    def try_take(self):
        """(ids, ticket); ([], -1) when nothing may start now; (None, -1) when the run is over"""
        n, ids, ticket = self._take(lib().pga_sched_try_take)
        if n > 0:
            return ids, ticket
        return ([], -1) if n == -2 else (None, -1)

    def finish(self, ticket: int) -> None:
        lib().pga_sched_finish(self.h, int(ticket))

    def abort(self) -> None:
        lib().pga_sched_abort(self.h)

    def left(self) -> int:
        return int(lib().pga_sched_left(self.h))


def cost(bases: int, n_seqs: int) -> float:
    return float(lib().pga_sched_cost(int(bases), int(n_seqs)))


def partition(pop, tasks: Sequence, world: int, per_rank: int = 4) -> List[int]:
    """owner per task from pga_sched_partition (the tree of a levels.Population: node 0 the root, children with larger ids)"""
    c0 = [nd.children[0] if nd.children else -1 for nd in pop.nodes]
    c1 = [nd.children[1] if nd.children else -1 for nd in pop.nodes]
    k = [_i32(c0), _i32(c1), _i32([t.node for t in tasks]), _i64([t.bases for t in tasks])]
    owner = np.zeros(max(1, len(tasks)), dtype=np.int32)
    rc = lib().pga_sched_partition(len(pop.nodes), k[0][1], k[1][1], len(tasks), k[2][1], k[3][1], int(world), int(per_rank), owner.ctypes.data_as(I32P))
    if rc < 0:
        raise ValueError(lib().pga_sched_error().decode())
    return [int(x) for x in owner[:len(tasks)]]


def run_ready_set(tasks: List, run_batch: Callable, slots: int = 6, cap_bases: float = 1.2e9, min_batch_bases: float = 0.0, done: Optional[set] = None,
                  only: Optional[set] = None, on_result: Optional[Callable] = None, express: int = 0, express_eps: float = 0.05, express_cap: float = 60e6):
    """schedule.run_ready_set with the library's scheduler: same arguments, same log"""
    ns = NativeSched(tasks)
    ns.start(slots, cap_bases, min_batch_bases, done, only, express, express_eps, express_cap)
    lock = threading.Lock()
    log, errs = [], []
    t_origin = time.perf_counter()

    def worker():
        while True:
            ids, ticket = ns.take()
            if ids is None:
                return
            t0 = time.perf_counter()
            try:
                res = run_batch([tasks[i] for i in ids])
                t1 = time.perf_counter()
                if on_result is not None:
                    on_result([tasks[i] for i in ids], res, t0 - t_origin, t1 - t_origin)
            except BaseException as e:   # noqa: BLE001
                with lock:
                    errs.append(e)
                ns.abort()
                return
            with lock:
                log.append((t0 - t_origin, t1 - t_origin, len(ids), sum(tasks[i].bases for i in ids)))
            ns.finish(ticket)

    th = [threading.Thread(target=worker, daemon=True) for _ in range(max(1, slots))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    ns.close()
    if errs:
        raise errs[0]
    return sorted(log)
