"""Routes that replaced rocPRIM sorts in round 6 (on by default), held against the rocPRIM routes of the same library on the same inputs -- every record, and
mid_occ itself.  What the records must BE is pinned, against the reference, by the other files of this suite through the default routes (i.e. through these).

  PGA_MAXOCC_HIST=0     mm_idx_cal_max_occ of every group by a sort of all keys instead of per-group histograms of the occurrence counts
                        (pangraph_amd/csrc/pga_maxocc_hist.h; logic under host emulation in tests/test_routes_emu.py)
  PGA_WG_SORT=0         the chaining stage's sorts of at most 4 096 pairs (segment lengths, chain candidates) by rocPRIM's block sort + merge passes instead of
                        one launch of one workgroup (pangraph_amd/csrc/pga_wg_sort.h; under emulation against std::stable_sort)

History: round 5 committed these (and a sort-free index, since removed: 3 % slower per build step) behind xfail markers without a device at hand, and the
driver's run recorded four expected failures.  Unmasked on an MI355X in round 6 every route passed alone; the failures were the subprocesses running out of
device memory beside the suite's own process, whose block cache held most of the device by then -- hence pga_trim() below before the first run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import json, os, sys
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from levels_util import product_align_groups, digest, high_occ_groups
from pangraph_amd.levels import Population
out = {}
waves = Population(9, 8, 200_000).build_waves()
for w in (0, 1, len(waves) - 2):                       # leaf pairs (two sequences of ~20 k minimizers per group), their second round, the root's first round (one group of many blocks)
    label, groups, names = waves[w]
    rows = product_align_groups(groups, names, sensitivity=10)
    out["wave %d" % w] = [[len(r), digest(r)] for r in rows]
groups, names = high_occ_groups()                      # k-mers that occur 520 and 4 200 times inside a group: long lists; the second overflows a bucket (sort route for that batch)
rows = product_align_groups(groups, names, sensitivity=10)
out["high_occ"] = [[len(r), digest(r)] for r in rows]
import ctypes, stagebind                              # mid_occ itself (options.c:70-76 over index.c:186-207), one group at a time, through the stage tap
dll = ctypes.CDLL(os.path.join(root, "pangraph_amd", "libpgalign.so"))
out["mid_occ"] = [[0, stagebind.product_chain(dll, g[:80], n[:80], sensitivity=10)[1]] for g, n in zip(groups, names)]
big = Population(4, 2, 3_000_000).build_waves()[0]     # one whole-genome-sized pair: ~600 k minimizers in one group, hundreds of buckets, dozens of tiles
rows = product_align_groups(big[1], big[2], sensitivity=10)
out["big pair"] = [[len(r), digest(r)] for r in rows]
print("RESULT " + json.dumps(out))
"""


def _run(extra_env, tail=""):
    # (the runs are processes of their own beside a pytest process whose block cache may hold most of the device by now: each keeps to a fifth of it)
    env = dict(os.environ, PGA_MEM_SHARE="0.2", **extra_env)
    r = subprocess.run([sys.executable, "-c", SCRIPT + tail, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


_BASE = {}


def _base():
    if not _BASE:
        # this process (the whole suite before this file) may hold most of the device in its block cache: give the idle blocks back before the runs beside it
        import ctypes
        dll = ctypes.CDLL(os.path.join(ROOT, "pangraph_amd", "libpgalign.so"))
        if hasattr(dll, "pga_trim"):
            dll.pga_trim()
        _BASE.update(_run({}))
        assert sum(n for v in _BASE.values() for n, _ in v) > 100
    return _BASE


def test_mid_occ_by_the_sort_gives_the_records_of_the_histograms():
    assert _run({"PGA_MAXOCC_HIST": "0"}) == _base()


def test_small_sorts_of_the_chaining_stage_by_rocprim_give_the_records_of_the_workgroup_sort():
    assert _run({"PGA_WG_SORT": "0"}) == _base()


def test_both_rocprim_routes_together():
    assert _run({"PGA_MAXOCC_HIST": "0", "PGA_WG_SORT": "0"}) == _base()


def test_a_cache_of_one_gigabyte_evicts_and_trims_without_changing_a_record():
    """pga_mem.cpp: with PGA_CACHE_GB=1 nearly every block a stage gives back is beyond the cache's limit -- the eviction loops of dev_alloc / dev_free run
    (oldest idle blocks of the current device first, at most 64 hipFree per miss) all through the run, and pga_trim() at the end releases what is left: the
    records are those of the default run, hipFree was really called, and after the trim the library holds (almost) nothing of the device."""
    tail = r"""
import ctypes
dll2 = ctypes.CDLL(os.path.join(root, "pangraph_amd", "libpgalign.so"))
st = (ctypes.c_int64 * 6)(); dll2.pga_mem_stats(st)
dll2.pga_trim.restype = ctypes.c_int64
freed = dll2.pga_trim()
st2 = (ctypes.c_int64 * 6)(); dll2.pga_mem_stats(st2)
print("MEM " + json.dumps({"hipFree_calls": int(st[2]), "trimmed": int(freed), "idle_after": int(st2[5])}))
"""
    env = dict(os.environ, PGA_MEM_SHARE="0.2", PGA_CACHE_GB="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT + tail, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])
    mem = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("MEM ")][-1][len("MEM "):])
    assert got == _base()
    assert mem["hipFree_calls"] > 0 and mem["idle_after"] == 0, mem
