"""Parity of the HIP path on the block sets of UPPER guide-tree levels and on the occurrence-driven seeding paths
(SURVEY.md section 8c fixtures (3), (4); configs C2, C3):
real block sets of finished builds (the reference's own test data), the sc2-like one-group population, groups whose
minimizers occur more often than mid_occ / max_max_occ, and every wave (both self-merge rounds of every merge of every tree
height) of a simulated build -- against golden digests generated from the reference build (tests/golden/make_golden_levels.py)
and, where oracle/_ref travels with the snapshot, against the reference run live on the box.  Also the run-length sort replay
through its own stage tap and concurrent mm_map from 16 threads.  Bit-exact."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from levels_util import ref_align_groups, product_align_groups, digest, high_occ_groups
from pangraph_amd.levels import Population, Rates, c2_population
from util import load_golden, read_fasta, rows_to_lists, GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def exp():
    return load_golden("levels_expected.json.gz")


def test_native_build_driver_equals_the_python_host(gpu_lib, tmp_path):
    """A whole build driven from the compiled host (pangraph_amd/host/build_driver.cpp: task file -> pga_sched_* -> six worker threads over
    pga_batch_create / pga_batch_align -> result file; no Python in that process) gives, call by call, the records the Python host gets for the same
    build: every field of every record and every CIGAR.  (What the records must BE is held against the reference elsewhere in this file.)
    First in the file: the driver is a process of its own and should find the device before this process's block cache has filled it."""
    import subprocess
    from conftest import ROOT
    from pangraph_amd import schedule as sched
    exe = os.path.join(ROOT, "pangraph_amd", "host", "build_driver")
    assert os.path.exists(exe), "built by make -C pangraph_amd/csrc"
    pop = Population(5, 12, 150_000)
    tasks = sched.build_tasks(pop)
    tf, of = str(tmp_path / "tasks.bin"), str(tmp_path / "out.bin")
    sched.write_task_file(tasks, tf, sensitivity=10, n_threads=8)
    r = subprocess.run([exe, tf, of, "6"], env=dict(os.environ, PGA_MEM_SHARE="0.3"), capture_output=True, text=True, timeout=600)   # a process of its own beside this one's block cache
    assert r.returncode == 0, r.stdout + r.stderr
    got, n_batches = sched.read_driver_results(of)
    assert len(got) == len(tasks) and n_batches >= 1
    results, _ = _run_build_ready_set(pop, tasks, 6, False, native=True)
    fields = [f for f in got[0][0].dtype.names if f not in ("group", "cigar_off", "pad")]
    seen, n_rec = set(), 0
    for ts, m, cg, _ in results:
        for g, t in enumerate(ts):
            mine = m[m["group"] == g]
            dm, dc = got[t.tid]
            assert len(mine) == len(dm), (t.tid, len(mine), len(dm))
            assert (dm["group"] == t.tid).all()
            for a, b in zip(mine, dm):
                assert all(a[f] == b[f] for f in fields), (t.tid, a, b)
                assert (cg[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["n_cigar"])] == dc[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["n_cigar"])]).all(), t.tid
            seen.add(t.tid); n_rec += len(dm)
    assert seen == set(range(len(tasks))) and n_rec > len(tasks)


def test_plasmid_block_set_real_ids(gpu_lib, exp):
    names, seqs = read_fasta(os.path.join(GOLDEN, "plasmids_blocks.fa.gz"))
    assert len(names) == 137
    assert rows_to_lists(gpu_lib.align_all(seqs, names, sensitivity=10)) == exp["plasmids_blocks"]["asm10"]
    for sens in (5, 20):
        rows = rows_to_lists(gpu_lib.align_all(seqs, names, sensitivity=sens))
        assert (len(rows), digest(rows)) == (exp["plasmids_blocks"][f"asm{sens}"]["n"], exp["plasmids_blocks"][f"asm{sens}"]["sha256"])


def test_staph_block_set(gpu_lib, exp):
    names, seqs = read_fasta(os.path.join(GOLDEN, "staph_blocks.fa.gz"))
    rows = product_align_groups([seqs], [names], sensitivity=10)[0]
    assert (len(rows), digest(rows)) == (exp["staph_blocks"]["n"], exp["staph_blocks"]["sha256"])


def test_russian_doll_plasmids(gpu_lib, exp):
    names, seqs = read_fasta(os.path.join(GOLDEN, "russian_doll_plasmids.fa.gz"))
    for sens in (10, 20):
        assert rows_to_lists(gpu_lib.align_all(seqs, names, sensitivity=sens)) == exp["russian_doll"][f"asm{sens}"]


def test_c2_sc2_like_group(gpu_lib, exp):
    seqs, names = c2_population()
    rows = product_align_groups([seqs], [names], sensitivity=10)[0]
    assert len(rows) == exp["c2"]["n"]
    assert digest(rows) == exp["c2"]["sha256"]


def test_high_occurrence_groups(gpu_lib, exp):
    groups, names = high_occ_groups()
    got = product_align_groups(groups, names, sensitivity=10)
    assert [dict(n=len(r), sha256=digest(r)) for r in got] == exp["high_occ"]


def test_c3_small_every_wave(gpu_lib, exp):
    p = exp["c3_small"]["params"]
    pop = Population(p["seed"], p["n"], p["length"], Rates(**p["rates"]))
    waves = pop.build_waves()
    assert len(waves) == len(exp["c3_small"]["waves"])
    for (label, groups, names), e in zip(waves, exp["c3_small"]["waves"]):
        assert [len(g) for g in groups] == e["n_blocks"], label
        got = product_align_groups(groups, names, sensitivity=10)
        assert [dict(n=len(r), sha256=digest(r)) for r in got] == e["groups"], label


@pytest.mark.parametrize("mode", ["check", "off"])
def test_c3_small_every_wave_length_bound_stop_checked_and_off(gpu_lib, exp, monkeypatch, mode):
    """The length-bound stop of end extensions (pga_dp.h) is on by default in the lane, pipeline and wave-strip kernels and ends a sweep on an analytic bound:
    the same multi-level build against the reference-made digests (a) with PGA_LB=check -- every problem the stop applies to stays with the lane kernel, which
    sweeps on behind the stop and fails the call if the record still changes -- and (b) with PGA_LB=off (the reference's full sweep everywhere).  Together with
    test_c3_small_every_wave (default: the stop taken unchecked) the three routes give the same records."""
    monkeypatch.setenv("PGA_LB", mode)
    p = exp["c3_small"]["params"]
    pop = Population(p["seed"], p["n"], p["length"], Rates(**p["rates"]))
    for (label, groups, names), e in zip(pop.build_waves(), exp["c3_small"]["waves"]):
        got = product_align_groups(groups, names, sensitivity=10)
        assert [dict(n=len(r), sha256=digest(r)) for r in got] == e["groups"], (mode, label)


@pytest.mark.parametrize("kw", [dict(sensitivity=5), dict(sensitivity=20), dict(sensitivity=10, kmer_length=15), dict(sensitivity=20, kmer_length=16),
                                dict(sensitivity=5, kmer_length=17)], ids=lambda k: "asm%d%s" % (k["sensitivity"], "-k%d" % k["kmer_length"] if "kmer_length" in k else ""))
def test_c3_small_every_wave_other_presets_live_vs_reference(gpu_lib, ref_lib, exp, kw):
    """the multi-level build of test_c3_small_every_wave under the other two presets and under -k (options.c:115-130; align_with_minimap2_lib.rs:42-57:
    sensitivity picks asm5 / asm10 / asm20, kmer_length overrides k; an even k takes the serial sketch kernel): every wave live against the compiled
    reference"""
    p = exp["c3_small"]["params"]
    pop = Population(p["seed"], p["n"], p["length"], Rates(**p["rates"]))
    n = 0
    for label, groups, names in pop.build_waves():
        got = product_align_groups(groups, names, **kw)
        want = ref_align_groups(groups, names, **kw)
        for g, (a, b) in enumerate(zip(got, want)):
            assert a == b, (label, g, len(a), len(b))
            n += len(a)
    assert n > 50


def test_c3_full_size_every_wave_vs_reference(gpu_lib, ref_lib):
    """config C3: 10 genomes x 5 Mbp, the whole build, every wave against the reference run here"""
    pop = Population(2, 10, 5_000_000)
    for label, groups, names in pop.build_waves():
        got = product_align_groups(groups, names, sensitivity=10)
        want = ref_align_groups(groups, names, sensitivity=10)
        for g, (a, b) in enumerate(zip(got, want)):
            assert a == b, (label, g, len(a), len(b))


def test_c5_first_three_heights_under_asm20_live_vs_reference(gpu_lib, ref_lib):
    """The BASELINE build's block sets under the OTHER scoring row (sensitivity 20 = asm20: k 19, w 10, a 1 b 4 q 6 e 2 q2 26 e2 1, options.c:125-130;
    the digests of the suite are asm10): every sixth group of both rounds of tree heights 1-3 (whole-genome pairs; a genome against a two-genome graph;
    block sets of a few dozen blocks) through the batch entry, live against the compiled reference on this box's cores.  asm20's w = 10 doubles the
    minimizers and anchors of a call and its low gap costs keep extensions alive longer: other DP classes, other band rings than under asm10."""
    waves = Population(20260928, 1000, 5_000_000).build_waves()[:6]
    n_groups = n_rec = 0
    for label, groups, names in waves:
        sel = list(range(0, len(groups), 6))
        gs, ns = [groups[i] for i in sel], [names[i] for i in sel]
        got = product_align_groups(gs, ns, sensitivity=20)
        want = ref_align_groups(gs, ns, sensitivity=20)
        for g, (a, b) in zip(sel, zip(got, want)):
            assert a == b, (label, g, len(a), len(b))
            n_rec += len(a)
        n_groups += len(sel)
    assert n_groups >= 100 and n_rec > 1000


def test_c5_every_wave(gpu_lib):
    """config C5 = the BASELINE configuration (1000 x 5 Mbp, seed 20260928): all 42 waves, all 1998 find_matches calls of the build
    through pga_batch_create + pga_batch_align, per group against the digests the compiled reference produced in the build container
    (tests/golden/make_golden_builds.py -> builds_expected.json.gz).  graph_merging.rs:95-128 is the call pattern."""
    e = load_golden("builds_expected.json.gz")["c5"]
    p = e["params"]
    waves = Population(p["seed"], p["n"], p["length"]).build_waves()
    assert len(waves) == len(e["waves"]) == 42
    n_groups = n_rec = 0
    for (label, groups, names), ew in zip(waves, e["waves"]):
        assert label == ew["label"] and [len(g) for g in groups] == ew["n_blocks"], label
        assert [int(sum(len(s) for s in g)) for g in groups] == ew["bases"], label
        got = product_align_groups(groups, names, sensitivity=10)
        bad = [g for g, (r, x) in enumerate(zip(got, ew["groups"])) if (len(r), digest(r)) != (x["n"], x["sha256"])]
        assert not bad, (label, bad[:5], len(bad))
        n_groups += len(groups)
        n_rec += sum(len(r) for r in got)
    assert n_groups == 1998 and n_rec == sum(x["n"] for w in e["waves"] for x in w["groups"])


def test_sort_replay_tap_adversarial(gpu_lib, ref_lib):
    """the run-length walk and the token walk of the sort replay (pga_sort_wave.h) against radix_sort_128x itself"""
    dll = gpu_lib.dll
    dll.pga_stage_sort.restype = C.c_int
    dll.pga_stage_sort.argtypes = [C.c_int32, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(77)
    arrays = []

    def add(x):
        arrays.append(np.asarray(x, dtype=np.uint64))

    for n in (1, 2, 63, 64, 65, 66, 1023, 1024, 1025, 4097, 70000, 400000):
        add(rng.integers(0, 5, size=n))                                          # few keys: long digit runs after the first level
        add(rng.integers(0, 1 << 40, size=n))                                    # (almost) distinct
        add(np.arange(n)[::-1] // 3)                                             # descending with ties
        add(np.roll(np.arange(n) // 7, n // 3))                                  # rotated ascending
        add((np.arange(n) // 50) << 16)                                          # ascending runs of 50 at byte level 2
        add(((np.arange(n)[::-1] // 300) << 16) | (rng.integers(0, 3, size=n)))  # descending runs + low-byte ties
        add(np.repeat(rng.integers(0, 1 << 32, size=max(1, n // 17)), 17)[:n])   # runs of 17 equal keys
        add(np.where(rng.random(n) < 0.02, rng.integers(0, 1 << 24, size=n), (np.arange(n) // 9) << 8))   # nearly sorted, 2 % foreign
        add(np.zeros(n))                                                         # one key
        add((rng.integers(0, 2, size=n) << 63) | (np.arange(n) // 40 << 8))      # two strands, each ascending
    seg_off = np.zeros(len(arrays) + 1, dtype=np.uint64)
    seg_off[1:] = np.cumsum([len(a) for a in arrays])
    xy = np.zeros((int(seg_off[-1]), 2), dtype=np.uint64)
    xy[:, 0] = np.concatenate(arrays)
    xy[:, 1] = np.arange(len(xy), dtype=np.uint64)
    want = xy.copy()
    for s in range(len(arrays)):
        b, e = int(seg_off[s]), int(seg_off[s + 1])
        ref_lib.dll.radix_sort_128x(C.c_void_p(want.ctypes.data + 16 * b), C.c_void_p(want.ctypes.data + 16 * e))
    assert dll.pga_stage_sort(len(arrays), seg_off.ctypes.data, xy.ctypes.data) == 0
    for s in range(len(arrays)):
        b, e = int(seg_off[s]), int(seg_off[s + 1])
        assert (xy[b:e] == want[b:e]).all(), (s, e - b)


def test_concurrent_mm_map_16_threads(gpu_lib, ref_lib):
    """what rayon does (align_with_minimap2_lib.rs:64-74): one index, mm_map from many threads, one mm_tbuf_t each"""
    from pangraph_amd.synth import evolve_population
    seqs = evolve_population(23, 48, 20000, snp=0.01, indel=0.001, n_inv=1, n_ins=1, n_del=1, max_event=3000)
    names = [str(1000 + 7919 * i) for i in range(len(seqs))]
    want = ref_lib.align_all(seqs, names, sensitivity=10)
    by_q = {}
    for r in want:
        by_q.setdefault(r.qname, []).append(r.key())
    io, mo = gpu_lib.make_options("asm10", s=90)
    idx = gpu_lib.index(seqs, names, io, mo)
    out, errs = {}, []

    def work(t):
        try:
            from pangraph_amd.mm2ffi import Mm2Index
            mine = Mm2Index(gpu_lib, idx.ptr, idx.io, idx.mo, None)      # own mm_tbuf_t on the shared index
            for q in range(t, len(seqs), 16):
                out[names[q]] = [r.key() for r in mine.map(seqs[q], names[q])]
            gpu_lib.dll.mm_tbuf_destroy(mine._tbuf)
            mine.ptr = None
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(16)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    idx.close()
    assert not errs, errs
    assert len(out) == len(seqs)
    for nm in names:
        assert out[nm] == by_q.get(nm, []), nm


def test_block_sequences_stay_resident_across_rounds(gpu_lib):
    """SURVEY 8(f)-4 / graph_merging.rs:26-69: the self-merge loop maps the merged graph again and again; most blocks of a round are the
    blocks of the round before.  One handle aligned twice gives the same list twice (nothing is re-uploaded), and the batch of the next round
    DERIVED from it -- kept blocks copied device to device out of the packed store, at new offsets, next to newly uploaded ones -- aligns
    exactly like a batch built from scratch from the same sequences."""
    from pangraph_amd import batch
    from pangraph_amd.synth import evolve_population, random_seq, mutate
    rng = np.random.default_rng(12)
    pop = evolve_population(31, 14, 30011, snp=0.01, indel=0.001, n_inv=1, n_ins=1, n_del=1, max_event=3000)       # odd lengths: nothing is word-aligned
    pop = [s if isinstance(s, str) else s.tobytes().decode() for s in pop]
    names = [str(70001 + 7919 * i) for i in range(len(pop))]
    g_old = [pop[0:5], pop[5:9], pop[9:14]]
    n_old = [names[0:5], names[5:9], names[9:14]]
    rb = batch.ResidentBatch(batch.PreparedBatch(g_old, n_old))
    first = [rows_to_lists(r) for r in rb.align(sensitivity=10, want_rows=True).groups]
    again = [rows_to_lists(r) for r in rb.align(sensitivity=10, want_rows=True).groups]
    assert first == again and sum(len(r) for r in first) > 10
    # the next round: blocks 1, 3, 4 of group 0 merged away into a new consensus, group 1 untouched, group 2 reordered with one new block
    new_a = mutate(rng, np.frombuffer(pop[1].encode(), dtype=np.uint8), snp=0.004, indel=0.0).tobytes().decode()
    new_b = random_seq(rng, 7777).tobytes().decode()
    g_new = [[0, new_a, 2], [5, 6, 7, 8], [13, 9, new_b, 11]]
    n_new = [[names[0], "424242", names[2]], names[5:9], [names[13], names[9], "99", names[11]]]
    derived = batch.DerivedBatch(rb, g_new, n_new)
    got = [rows_to_lists(r) for r in derived.align(sensitivity=10, want_rows=True).groups]
    flat_old = [s for g in g_old for s in g]
    g_ref = [[flat_old[s] if isinstance(s, int) else s for s in g] for g in g_new]
    want = product_align_groups(g_ref, n_new, sensitivity=10)
    assert got == want and sum(len(r) for r in got) > 5
    rb.close()                                                           # the old batch may go: the derived one owns its copy
    assert [rows_to_lists(r) for r in derived.align(sensitivity=10, want_rows=True).groups] == want


@pytest.mark.gpu
def test_calls_take_their_inputs_from_a_resident_library(gpu_lib):
    """bench.py --inputs resident: the block sequences of EVERY find_matches call of a build are handed over once (one batch handle, never
    aligned); a batch of calls is then derived from it -- every sequence by a device-to-device copy of its packed words, nothing crosses PCIe
    -- and must align exactly like the same calls handed over as host strings."""
    from pangraph_amd import batch
    from pangraph_amd import schedule as sched
    pop = Population(5, 6, 60_013)                                       # odd length: the calls' sequences start at every word phase
    tasks = sched.build_tasks(pop)
    first, n = {}, 0
    for t in tasks:
        first[t.tid] = n
        n += len(t.seqs)
    lib = batch.ResidentBatch(sched.TaskBatch(tasks))
    picks = [tasks[::2], tasks[1::2], tasks[-3:]]                        # batches that take their calls from all over the library
    total = 0
    for ts in picks:
        host = batch.ResidentBatch(sched.TaskBatch(ts))
        want = host.align(sensitivity=10, want_raw=True)
        dev = batch.ResidentBatch(sched.TaskBatch(ts, first), derive_from=lib)
        got = dev.align(sensitivity=10, want_raw=True)
        assert np.array_equal(np.asarray(got.raw_matches), np.asarray(want.raw_matches))
        assert np.array_equal(np.asarray(got.raw_cigars), np.asarray(want.raw_cigars))
        total += int(want.stats["n_matches"])
        for r in (want, got):
            r.close()
        host.close(); dev.close()
    assert total > 20
    lib.close()


@pytest.mark.gpu
def test_busy_interval_log_and_stream_pool(gpu_lib):
    """Measurement helpers of include/pga_align.h: between pga_busy_begin and pga_busy_end every event-bracketed launch leaves its interval;
    the union per kernel family cannot exceed the union over all families, nor the sum of the family's device times.  pga_warm_streams
    fills the pool the batch handles lease their streams from."""
    from pangraph_amd import batch
    assert batch.lib().pga_warm_streams(3) == 0
    pop = Population(7, 4, 40_000)
    waves = pop.build_waves()
    _, groups, names = waves[0]
    pb = batch.PreparedBatch(groups, names)
    batch.busy_begin()
    rb = batch.ResidentBatch(pb)
    res = rb.align(sensitivity=10, want_raw=True)
    st = res.stats
    busy = batch.busy_end()
    assert busy["intervals"] > 5 and busy["any"] > 0.0
    for i, name in enumerate(batch.KERNELS):
        if name in busy:
            assert busy[name] <= busy["any"] * 1.001 + 1e-3
            assert busy[name] <= st["kern_ms"][i] * 1.001 + 1e-3, name      # one batch, launches of a family do not overlap themselves much
    res.close(); rb.close()
    assert batch.busy_end()["intervals"] == 0                             # closed: nothing is logged any more


def _run_build_ready_set(pop, tasks, slots, resident, native=False):
    """the benchmark's execution mode (bench.py:step_ready): the find_matches calls of the build under the ready-set schedule, `slots` batches in
    flight from `slots` host threads, inputs handed over as host strings (pga_batch_create) or derived from a resident library (pga_batch_derive)"""
    from pangraph_amd import batch
    from pangraph_amd import schedule as sched
    from pangraph_amd.dist import MATCH_DTYPE
    lib, first = None, None
    if resident:
        first, n = {}, 0
        for t in tasks:
            first[t.tid] = n
            n += len(t.seqs)
        lib = batch.ResidentBatch(sched.TaskBatch(tasks))
    results, lock = [], threading.Lock()

    def run_batch(ts):
        batch.set_device(0)
        rb = batch.ResidentBatch(sched.TaskBatch(ts, first), derive_from=lib)
        res = rb.align(sensitivity=10, want_raw=True, n_threads=8)
        rb.close()
        return res

    def on_result(ts, res, t0, t1):
        m = np.array(res.raw_matches, copy=True).view(MATCH_DTYPE)
        cg = np.array(res.raw_cigars, copy=True).view(np.uint32)
        res.close()
        with lock:
            results.append((ts, m, cg, None))

    log = sched.run_ready_set(tasks, run_batch, slots=slots, cap_bases=1.2e9, on_result=on_result, native=native)
    if lib is not None:
        lib.close()
    return results, log


@pytest.mark.parametrize("resident", [False, True], ids=["inputs_host", "inputs_resident"])
def test_c5_ready_set_six_slots_every_call_vs_reference_digests(gpu_lib, resident):
    """The execution mode bench.py times -- NOT level-synchronous waves: the 1998 find_matches calls of the BASELINE build (config C5) in
    dependency order (graph_merging.rs:26-69, build_run.rs:111-128: round 0 of a merge needs the last round of both children, round r needs
    round r - 1), SIX batches in flight from six host threads (own stream, own arena, shared DP lane pools), inputs as host strings and, second
    run, device-to-device out of a resident library.  Every call against the digest the compiled reference produced for it.  The first run takes its
    batches from schedule.ReadySet, the second from the library's scheduler (pga_sched_*: what bench.py and a Rust host are driven by)."""
    from pangraph_amd import batch
    from pangraph_amd import digest as dg
    from pangraph_amd import schedule as sched
    gold = load_golden("builds_expected.json.gz")["c5"]
    p = gold["params"]
    pop = Population(p["seed"], p["n"], p["length"])
    tasks = sched.build_tasks(pop)
    assert len(tasks) == 1998
    want = dg.expected_by_call(pop, gold)
    batch.lib().pga_warm_streams(6)
    # what the device holds when this test starts (the tests before it leave their blocks in the library's cache), and the library's / the runtime's own
    # stderr of the run in a file of its own: an abort inside the runtime takes pytest's captured output with it
    import ctypes
    import json
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    tag = "resident" if resident else "host"
    a = (ctypes.c_int64 * 6)()
    batch.lib().pga_mem_stats(a)
    with open(os.path.join(out_dir, f"sixslot_mem_{tag}.json"), "w") as fh:
        json.dump({"hipMalloc_calls": a[0], "hipFree_calls": a[2], "live_GB": a[4] / 2**30, "idle_in_cache_GB": a[5] / 2**30}, fh)
    saved = os.dup(2)
    fd = os.open(os.path.join(out_dir, f"sixslot_stderr_{tag}.txt"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.dup2(fd, 2)
    try:
        results, log = _run_build_ready_set(pop, tasks, 6, resident, native=resident)
    finally:
        os.dup2(saved, 2); os.close(saved); os.close(fd)
    assert max(sum(1 for a, b, _, _ in log if a <= t < b) for t, _, _, _ in log) >= 4          # batches really were in flight together
    n, bad = dg.check_calls(results, want)
    assert n == 1998 and not bad, (n, bad[:10], len(bad))
    assert sum(len(m) for _, m, _, _ in results) == sum(x["n"] for w in gold["waves"] for x in w["groups"])
