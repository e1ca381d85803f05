#!/usr/bin/env python3
"""bench.py -- aligned Gbp/s of the pairwise block-alignment backend over a WHOLE `pangraph build` of 1000 synthetic
5 Mbp genomes (BASELINE.json metric and configuration C5).

Workload: `pangraph_amd.levels.Population(20260928, 1000, 5_000_000)` -- genomes evolved along a random tree with SNPs,
indels, inversions, HGT insertions, deletions and duplications -- and, for every merge of the guide tree, the block sets the
reference would hand to its aligner: round 0 (joined child graphs) and round 1 (merged graph) of the self-merge loop
(graph_merging.rs:26-69).  A WAVE is what a level-synchronous host can align at once: all merges of one tree height, one round.
A STEP is the whole build: every wave in dependency order, each through `pga_batch_create` (host hand-over: upload + encoding)
and `pga_batch_align` (sketch -> index -> seed -> chain -> extend -> records), followed -- for N > 1 -- by the RCCL gather of
the wave's match list to rank 0.  Units U = bases handed to the aligner, summed over all waves (every sequence of every
`find_matches` call counts once); value = U * steps / wall time INCLUDING the hand-over of the sequences
(8d's definition); `resident_gbp_s` is the same without the hand-over (bases already in HBM).

N > 1: the FIXED workload is sharded (strong scaling): the groups of every wave are dealt to the ranks balanced by base count,
no data-path collective, one match-list gather per wave.  `python bench.py --gpus N` spawns the N ranks itself when it is not
already running under torch.distributed.run.

The JSON line also carries
  roofline      the kernel with the largest device time and a per-kernel table: algorithmic bytes / HIP-event time vs 8 TB/s for the
                HBM-bound kernels, evaluated DP cells per second for the DP kernels
  cpu_baseline  the REFERENCE's own C (oracle/_ref/libmm2ref.so, compiled from /root/reference by oracle/Makefile), one process
                per host core, on a bounded sample of the same waves
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import pangraph_amd  # noqa: E402  (sets the HIP runtime defaults of the backend before anything initialises HIP)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def usable_cpus() -> int:
    """CPUs this process can really use: affinity mask capped by the cgroup v2 quota (the GPU box shows 256 but grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pmc_traffic(kernel: str, launches_per_step: float):
    """HBM bytes of `kernel` per LAUNCH AS THE ROOFLINE COUNTS LAUNCHES (launches_per_step: e.g. one sort replay = many dispatches), from
    the committed rocprofv3 --pmc passes of this same workload (profiles/r*_pmc_hbm_traffic_c5.json: one step per pass, FETCH_SIZE and
    WRITE_SIZE in separate passes, in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950).  The same unit as
    roofline.alg_bytes_per_launch.  NOT measured in this run; None if there is no PMC summary."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic_c5.json")))
    if not files or not launches_per_step:
        return None
    d = json.load(open(files[-1]))
    steps = float(d.get("_meta", {}).get("steps", 1))
    names = [x for x in kernel.replace(" ", "").split("+") if x.startswith("k_")]
    if not names:
        return None
    tot = 0.0
    for c, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        for k, v in d.get(c, {}).items():
            if any(nm in k for nm in names):
                tot += mul * v["sum"] * 1024.0
    return tot / steps / launches_per_step if tot else None


def pmc_issue():
    """VALU / LDS issue fractions of the DP kernels (SQ_ACTIVE_INST_VALU|LDS / SQ_WAVE_CYCLES) from the committed rocprofv3 --pmc passes of this
    same workload (profiles/r*_pmc_issue*kernels.json, dev/gpu.sh profile <tag> issue).  NOT measured in this run; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_issue*kernels.json")))
    if not files:
        return None
    d = json.load(open(files[-1])).get("kernels", {})
    return {"source": os.path.relpath(files[-1], ROOT),
            "kernels": {k.replace("pga::", ""): {"valu_issue_frac": v.get("valu_issue_frac_of_wave_cycles"), "lds_issue_frac": v.get("lds_issue_frac_of_wave_cycles")}
                        for k, v in d.items() if any(t in k for t in ("k_extd2", "k_gapfill", "k_ll_i16", "k_approx_strips", "k_wstrips", "k_bstrips"))}}


# Cycles ONE wave spends on the cell recurrence per cell it evaluates, measured inside the kernels (dev builds with cycle counters): what the kernel's
# own arithmetic allows if nothing else cost anything.  k_extd2_lanes: 1 370 cycles of a 5 170-cycle diagonal for the 512 cells of a wave
# (-DPGA_LANES_PROF); k_ext_pipe: the same packed arithmetic, four columns per lane (dev/pipe_prof.sh: 2 450-cycle diagonal of 256 cells,
# the cell block a third of its instructions).  The other DP families use the lane kernel's figure.
DP_CELL_CYCLES = {"k_extd2_lanes": 1370.0 / 512.0, "k_ext_pipe": 820.0 / 256.0}
N_SIMD, CLOCK_HZ = 1024, 2.4e9   # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock


VALU_PEAK_WAVE_INST_S = N_SIMD * CLOCK_HZ / 2.0   # MI355X_MICROARCH.md "Wave scheduling": a SIMD-32 issues one wave64 VALU instruction over 2 cycles


def _short(k: str) -> str:
    return k.replace("void ", "").replace("pga::", "").split("(")[0]


def pmc_valu_insts(names):
    """VALU wave-instructions per STEP of the kernels whose names contain one of `names`, from the committed rocprofv3 --pmc pass of this same workload
    (profiles/r*_pmc_issue*kernels.json: SQ_INSTS_VALU summed over all dispatches of one step).  NOT measured in this run; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_issue*kernels.json")))
    if not files:
        return None
    tot = 0.0
    for k, v in json.load(open(files[-1])).get("kernels", {}).items():
        if any(nm in k for nm in names):
            tot += float(v.get("SQ_INSTS_VALU", 0.0))
    return tot or None


def rocprof_top_kernel():
    """The kernel with the largest total duration in the committed `rocprofv3 --kernel-trace --stats` summary of this same command
    (profiles/r*_c5_kernel_stats.csv: warm-up step + one timed step in the file), with BOTH fractions for it: VALU issue (SQ_INSTS_VALU of the PMC pass
    over its rocprof time, against 1024 SIMDs x 2.4 GHz / 2) and HBM (counter bytes FETCH x 2 + WRITE over its rocprof time, against 8 TB/s).
    HIP events on a stream see a launch from the moment it is queued behind the stream's earlier work; rocprof sees the kernel itself -- the two rankings
    can differ, hence both in the line.  NOT measured in this run; None if there is no summary."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c5_kernel_stats.csv")))
    if not files:
        return None
    rows = []
    with open(files[-1], newline="") as fh:
        for r in csv.DictReader(fh):
            try:
                rows.append((_short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["MaxNs"])))
            except (KeyError, ValueError):
                continue
    rows = [r for r in rows if r[0].startswith("k_")]
    if not rows:
        return None
    steps_in_file = 2.0
    name, calls, tot_ns, avg_ns, max_ns = max(rows, key=lambda r: r[2])
    fam = name.split("<")[0]
    out = {"name": name, "calls_per_step": calls / steps_in_file, "avg_ms": avg_ns * 1e-6, "max_ms": max_ns * 1e-6, "kernel_ms_per_step": tot_ns * 1e-6 / steps_in_file,
           "source": os.path.relpath(files[-1], ROOT)}
    secs = tot_ns * 1e-9 / steps_in_file
    vi = pmc_valu_insts([name if "<" in name else fam + "("]) or pmc_valu_insts([fam])
    if vi and secs > 0:
        out["valu_wave_insts_per_s"] = vi / secs
        out["valu_frac"] = vi / secs / VALU_PEAK_WAVE_INST_S
    tr = pmc_traffic(fam, 1.0)            # bytes per step
    if tr and secs > 0:
        out["hbm_counter_GBs"] = tr / secs / 1e9
        out["hbm_frac"] = tr / secs / 1e9 / HBM_PEAK_GBS
    return out


def profile_counts(ms_step: float):
    """From the committed rocprofv3 summaries of this same workload (profiles/r*_kernel_concurrency.json: kernel trace of warm-up + one step;
    profiles/r*_pmc_issue*kernels.json: SQ_WAVE_CYCLES of one step, in quad-cycles): kernel dispatches per step and the average number of waves
    resident per SIMD over a step of ms_step.  NOT measured in this run; None where there is no summary."""
    import glob
    out = {"dispatches_per_step": None, "resident_waves_per_simd": None}
    f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_concurrency.json")))
    if f:
        d = json.load(open(f[-1]))
        out["dispatches_per_step"] = int(d.get("kernels", 0) / float(d.get("steps_in_trace", 2)))
        out["dispatches_source"] = os.path.relpath(f[-1], ROOT)
    f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_issue*kernels.json")))
    if f and ms_step > 0:
        d = json.load(open(f[-1]))
        wc = d.get("all_kernels_SQ_WAVE_CYCLES")
        if wc is None:
            wc = sum(v.get("SQ_WAVE_CYCLES", 0.0) for v in d.get("kernels", {}).values())
        out["resident_waves_per_simd"] = round(4.0 * wc / (ms_step * 1e-3 * CLOCK_HZ * N_SIMD), 3)
        out["waves_source"] = os.path.relpath(f[-1], ROOT)
    return out


def _cpu_worker(args):
    so, seqs, names = args
    from pangraph_amd.mm2ffi import Mm2Lib
    lib = Mm2Lib(so)
    t0 = time.time()
    rows = lib.align_all([s.tobytes().decode() for s in seqs], names, sensitivity=10)
    return time.time() - t0, sum(len(s) for s in seqs), len(rows)


def cpu_baseline(waves, budget_s: float):
    """The reference C on the host cores, on WHOLE waves of the same build: the leaf wave (height 1, round 0), a wave from the middle of
    the tree and the root's round 0 -- every group of each, one group per process, all usable cores busy (largest group first); then the
    same code on ONE core for a leaf pair and a mid-tree group.  A wave whose estimated cost exceeds the budget is sampled (every k-th
    group) and the sample says so."""
    import multiprocessing as mp
    so = os.path.join(ROOT, "oracle", "_ref", "libmm2ref.so")
    kind = "reference"
    if not os.path.exists(so):
        so, kind = os.path.join(ROOT, "oracle", "libpgoracle.so"), "port"
    cores = usable_cpus()
    r0 = [i for i, (label, _, _) in enumerate(waves) if "round 0" in label] or [0]
    picks = sorted({r0[0], r0[len(r0) // 2], r0[-1]})
    # calibrate on the smallest group of the first wave
    _, g0, n0 = waves[picks[0]]
    i0 = min(range(len(g0)), key=lambda i: sum(len(s) for s in g0[i]))
    t1, b1, _ = _cpu_worker((so, g0[i0], n0[i0]))
    rate1 = b1 / max(t1, 1e-3)
    jobs, notes = [], []
    for w in picks:
        label, groups, names = waves[w]
        sizes = [sum(len(s) for s in g) for g in groups]
        est = sum(sizes) / rate1 / cores
        step = max(1, int(-(-est // max(budget_s, 1e-3))))                 # every step-th group keeps the wave inside the budget
        ids = list(range(0, len(groups), step))
        jobs += [(so, groups[i], names[i]) for i in ids]
        notes.append(f"{label.split(' (')[0]}: {len(ids)} of {len(groups)} groups" + ("" if step == 1 else f" (every {step}th)"))
    n = min(cores, len(jobs))
    jobs.sort(key=lambda j: -sum(len(s) for s in j[1]))
    t0 = time.time()
    with mp.get_context("spawn").Pool(n) as pool:
        res = pool.map(_cpu_worker, jobs, chunksize=1)
    wall = time.time() - t0
    bases = sum(r[1] for r in res)
    # one core: a leaf pair (above) and the median group of the middle wave
    _, gm, nm = waves[picks[len(picks) // 2]]
    im = sorted(range(len(gm)), key=lambda i: sum(len(s) for s in gm[i]))[len(gm) // 2]
    tm, bm, _ = _cpu_worker((so, gm[im], nm[im]))
    return {"value": bases / wall / 1e9, "unit": "Gbp/s", "cores": n, "kind": kind,
            "sample": f"whole waves, every group: {'; '.join(notes)} ({bases / 1e6:.1f} Mbp), {n} processes x 1 thread, {wall:.1f} s wall, "
                      f"{sum(r[0] for r in res):.1f} core-s",
            "one_core_gbp_s": {"leaf_pair": rate1 / 1e9, "mid_tree_group": bm / max(tm, 1e-3) / 1e9}}


def respawn_under_torchrun(n: int):
    port = 29500 + os.getpid() % 20000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port)] + sys.argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def next_rows(pop, seed: int):
    """SURVEY 8(f) rows that are built, timed once each OUTSIDE the timed region (rank 0, N = 1): the guide tree of the population's leaf
    genomes and the re-alignment of member sequences onto block consensus sequences (synthetic merge: 200 blocks of ~10 kb, 100 members
    each).  Reported next to the headline, never part of `value`.  A failure is reported as text, it does not fail the bench."""
    import ctypes as C
    import numpy as np
    from pangraph_amd import batch
    out = {}
    dll = batch.lib()
    try:
        genomes = [pop.genomes[v] for v in pop.leaves]
        n = len(genomes)
        ptrs = (C.c_char_p * n)(*[C.cast(g.ctypes.data, C.c_char_p) for g in genomes])
        lens = (C.c_uint32 * n)(*[len(g) for g in genomes])
        merges = np.zeros((max(n - 1, 1), 2), dtype=np.int32)
        dll.pga_guide_tree.restype = C.c_int
        dll.pga_guide_tree.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        t0 = time.perf_counter()
        rc = dll.pga_guide_tree(n, ptrs, lens, 15, 100, None, merges.ctypes.data)
        dt = time.perf_counter() - t0
        gbp = sum(len(g) for g in genomes) * 1e-9
        out["f3_guide_tree"] = {"genomes": n, "Gbp": gbp, "seconds": dt, "gbp_s": gbp / dt, "rc": rc, "entry": "pga_guide_tree (mash distance k=15 w=100 + neighbor joining)"}
    except Exception as e:                                       # noqa: BLE001
        out["f3_guide_tree"] = {"error": repr(e)}
    try:
        from pangraph_amd import mapvar as mb
        rng = np.random.default_rng(seed)
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        refs, qrys = [], []
        for b in range(200):
            ref = acgt[rng.integers(0, 4, int(10000 * rng.uniform(0.5, 1.5)))]
            rb = ref.tobytes()
            for m in range(100):
                a = ref.copy()
                k = rng.random(len(a)) < 0.01
                a[k] = acgt[rng.integers(0, 4, int(k.sum()))]
                q = a.tobytes()
                for _ in range(int(rng.integers(0, 4))):
                    p0 = int(rng.integers(0, len(q))); ln = int(rng.integers(1, 20))
                    q = q[:p0] + q[p0 + ln:] if rng.random() < 0.5 else q[:p0] + acgt[rng.integers(0, 4, ln)].tobytes() + q[p0:]
                refs.append(rb); qrys.append(q)
        n = len(qrys)
        J = (mb.job_t * n)()
        R = (mb.res_t * n)()
        subs = C.POINTER(mb.sub_t)(); dels = C.POINTER(mb.del_t)(); inss = C.POINTER(mb.ins_t)(); iseq = C.POINTER(C.c_char)()
        dll.pga_map_variations.restype = C.c_int
        dll.pga_map_variations.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        dll.pga_free.argtypes = [C.c_void_p]
        p = mb.params()
        mbp = sum(len(q) for q in qrys) * 1e-6
        runs = []
        for band in (20, 2):                                      # 2 + 5: bands of 15 columns, four jobs per wave (k_mapvar_packed)
            for i in range(n):
                J[i].ref = refs[i]; J[i].qry = qrys[i]; J[i].ref_len = len(refs[i]); J[i].qry_len = len(qrys[i]); J[i].mean_shift = 0; J[i].band_width = band
            best = None
            for rep in range(2):                                  # the first call also grows the allocator's pools
                t0 = time.perf_counter()
                rc = dll.pga_map_variations(n, J, C.byref(p), R, C.byref(subs), C.byref(dels), C.byref(inss), C.byref(iseq))
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
                for ptr in (subs, dels, inss, iseq):
                    if ptr:
                        dll.pga_free(C.cast(ptr, C.c_void_p))
            # (the round trip through Edit::apply and the comparison with the restatement are the GPU tests' job; here: every job finished)
            runs.append({"band_width": band, "seconds": best, "gbp_s": mbp * 1e-3 / best, "rc": rc, "sample_status_ok": all(R[i].status == 0 for i in range(0, n, 97)),
                         "retried": sum(R[i].attempts > 1 for i in range(n))})
        out["f1_map_variations"] = {"members": n, "member_Mbp": mbp, "seconds": runs[0]["seconds"], "gbp_s": runs[0]["gbp_s"], "runs": runs,
                                    "entry": "pga_map_variations (band + 5, up to 4 attempts), upload and edit download included"}
    except Exception as e:                                       # noqa: BLE001
        out["f1_map_variations"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=int(os.environ.get("PGA_BENCH_GENOMES", 1000)), help="genomes of the whole build (not per GPU)")
    ap.add_argument("--length", type=int, default=int(os.environ.get("PGA_BENCH_LENGTH", 5_000_000)))
    ap.add_argument("--seed", type=int, default=20260928)
    ap.add_argument("--leaf-only", action="store_true", help="diagnosis: only the waves of height 1")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU-baseline work per core (0 disables)")
    ap.add_argument("--schedule", choices=("ready", "waves"), default=os.environ.get("PGA_BENCH_SCHEDULE", "ready"),
                    help="ready: a find_matches call starts when the calls it depends on are done (pangraph_amd/schedule.py); waves: level-synchronous, one batch per wave")
    ap.add_argument("--inputs", choices=("resident", "host"), default=os.environ.get("PGA_BENCH_INPUTS", "host"),
                    help="host (default, SURVEY 8d): every call hands over host strings inside the timed region (H2D + encoding are timed); resident: the block "
                         "sequences of every call are in HBM (packed store) before the timed region, a call takes its inputs by a device-to-device copy "
                         "(pga_batch_derive) -- reported as the secondary `resident_gbp_s` by a default run")
    ap.add_argument("--detail", default=os.environ.get("PGA_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")),
                    help="where the long form goes (per-kernel table, batch timeline, stage sums, next rows); the LAST line of stdout stays under 4 KB")
    ap.add_argument("--no-resident-rate", action="store_true", help="skip the one extra step with resident inputs after the timed region (N = 1)")
    ap.add_argument("--no-parity-check", action="store_true", help="do not digest the last timed step's records against tests/golden/builds_expected.json.gz")
    ap.add_argument("--model-gbp-s", type=float, default=7.9, help="N > 1: single-GPU rate the scaling model is evaluated with (profiles/r05_d_bench_c5.json)")
    ap.add_argument("--model-host-s-per-gbp", type=float, default=0.85, help="N > 1: host core-seconds per Gbp of the single-GPU run (14.5 s per 17.07 Gbp step)")
    ap.add_argument("--slots", type=int, default=int(os.environ.get("PGA_BENCH_SLOTS", 6)), help="batches in flight (ready-set schedule)")
    ap.add_argument("--express", type=int, default=int(os.environ.get("PGA_BENCH_EXPRESS", 0)), help="slots reserved for the calls on the longest remaining path (schedule.run_ready_set: express)")
    ap.add_argument("--cap-gbp", type=float, default=float(os.environ.get("PGA_BENCH_CAP_GBP", 1.2)), help="largest batch of the ready-set schedule")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the one-off timings of the SURVEY 8(f) rows (guide tree, map_variations) reported next to the headline")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args.gpus)
    # the interpreter hands its lock from a thread that does not release it to a waiting one after this many seconds (CPython's default: 5 ms); the six
    # driving threads wait for it whenever a batch ends or starts, and the calls of the upper tree take 5-15 ms each
    if os.environ.get("PGA_BENCH_SWITCH_INTERVAL"):
        sys.setswitchinterval(float(os.environ["PGA_BENCH_SWITCH_INTERVAL"]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist
    from pangraph_amd import batch
    from pangraph_amd.dist import gather_matches, max_over_ranks, shard_groups_balanced
    from pangraph_amd.levels import Population, waves_bases

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    # PGA_BENCH_SINGLE_DEVICE=1 (debugging on a 1-GPU box): every rank computes on GPU 0 and the collectives run over gloo
    single = os.environ.get("PGA_BENCH_SINGLE_DEVICE") == "1"
    if single:
        local = 0
        os.environ.setdefault("PGA_MEM_SHARE", str(1.0 / world))      # the ranks share one device's memory: each caches and reserves its part only
        args.slots = max(1, args.slots // world)
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local}, the node exposes {torch.cuda.device_count()} (one rank per GPU; PGA_BENCH_SINGLE_DEVICE=1 puts every rank on GPU 0 for debugging)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    batch.set_device(local)
    if world > 1:
        if single:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cdev = torch.device("cpu") if single else dev          # where the collectives' tensors live

    t_gen = time.time()
    pop = Population(args.seed, args.genomes, args.length)
    waves = pop.build_waves()
    if args.leaf_only:
        waves = waves[:2]
    units = float(waves_bases(waves))
    t_gen = time.time() - t_gen

    # the sharding plan of every wave (identical on every rank) and this rank's flat C views of its groups
    plans, mine = [], []
    for _, groups, names in waves:
        plan = shard_groups_balanced([sum(len(s) for s in g) for g in groups], world)
        plans.append(plan)
        ids = plan[rank]
        mine.append(batch.PreparedBatch([groups[i] for i in ids], [names[i] for i in ids]) if ids else None)

    # host threads of this rank: the ranks of a node share one CPU quota
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # (most of a batch's helper threads wait for the device: with eight ranks on a 16-core grant a rank still gets four -- the grant is a CPU-time quota,
    # not a thread limit -- and the model in schedule.predict_scaling prices the host's share)
    n_threads = max(4, usable_cpus() // max(1, local_world))

    def add_stats(agg, st):
        if agg["stats"] is None:
            agg["stats"] = {k: (list(v) if isinstance(v, list) else v) for k, v in st.items()}
        else:
            for k, v in st.items():
                if isinstance(v, list):
                    agg["stats"][k] = [a + b for a, b in zip(agg["stats"][k], v)]
                else:
                    agg["stats"][k] += v

    # ---- ready-set schedule: the find_matches calls of the build with their true dependencies -------------------------------
    import threading
    from pangraph_amd import schedule as sched
    tasks = sched.build_tasks(pop) if not args.leaf_only else [t for t in sched.build_tasks(pop) if pop.nodes[t.node].height == 1]
    if args.leaf_only:
        for i, t in enumerate(tasks):
            t.deps = [] if t.round == 0 else [i - 1]
            t.tid = i
        sched.finish(tasks)
    owner, _ = sched.partition_subtrees(pop, tasks, world) if world > 1 and not args.leaf_only else ([0] * len(tasks), None)
    for t in tasks:
        if owner[t.tid] in (rank, -1) or world == 1:
            t.prepare()
    if os.environ.get("PGA_BENCH_WARM_STREAMS", "1") != "0":
        batch.lib().pga_warm_streams(int(args.slots))
    slot_threads = int(os.environ.get("PGA_BENCH_SLOT_THREADS", max(2, n_threads // max(1, min(args.slots, 2)))))
    # resident inputs: one batch handle holds the sequences of every call this rank can meet (0.375 B per base in HBM);
    # inp["first"][tid] = index of the call's first sequence in it.  Only the ready-set schedule can take its inputs from there.
    inp = {"lib": None, "first": None, "t_lib": None}

    def make_resident():
        t_lib = time.time()
        own = [t for t in tasks if owner[t.tid] in (rank, -1) or world == 1]
        first, n_lib = {}, 0
        for t in own:
            first[t.tid] = n_lib
            n_lib += len(t.seqs)
        inp["lib"], inp["first"] = batch.ResidentBatch(sched.TaskBatch(own)), first
        inp["t_lib"] = time.time() - t_lib

    if args.inputs == "resident" and args.schedule == "ready":
        make_resident()
    # the secondary rate (resident inputs, after the timed region) takes its library NOW: 6.4 GB of HBM that the timed steps do not touch.  Made
    # after them, its construction (17 GB of temporaries) lands on a cache the steps have filled and the allocator starts giving blocks back --
    # hipFree synchronises the device and the following steps ran 1.6x slower (measured; pga_mem_stats shows it)
    stash = None
    want_secondary = world == 1 and args.schedule == "ready" and inp["lib"] is None and not args.no_resident_rate and not args.leaf_only
    if want_secondary:
        make_resident(); stash = dict(inp); inp.update({"lib": None, "first": None})

    def mem_stats():
        import ctypes
        a = (ctypes.c_int64 * 6)()
        batch.lib().pga_mem_stats(a)
        return {"hipMalloc_calls": a[0], "hipMalloc_s": a[1] * 1e-9, "hipFree_calls": a[2], "hipFree_s": a[3] * 1e-9, "live_GB": a[4] / 2**30, "idle_in_cache_GB": a[5] / 2**30}

    # who takes the schedule's decisions: the library (pga_sched_*, what a Rust host binds; the default) or schedule.ReadySet (PGA_NATIVE_SCHED=0).
    # Same batches either way (tests/test_schedule_cpu.py); medians of six steps, ABAB on one box: 2 013 / 2 007 against 2 077 / 2 028 ms
    native_sched = os.environ.get("PGA_NATIVE_SCHED", "1") not in ("", "0")

    def step_ready():
        from pangraph_amd.dist import MATCH_DTYPE, gather_blobs, merge_match_lists
        agg = {"create_s": 0.0, "align_s": 0.0, "gather_s": 0.0, "n_matches": 0, "stats": None, "per_wave": [], "batches": [], "results": []}
        lock = threading.Lock()
        recs, pools, pool_len = [], [], [0]

        def run_batch(ts):
            tp0 = time.perf_counter()
            batch.set_device(local)                               # HIP's current device belongs to the host thread: every worker says which one it means
            tb = sched.TaskBatch(ts, inp["first"])
            t0 = time.perf_counter()
            with lock:
                agg["py_s"] = agg.get("py_s", 0.0) + (t0 - tp0)
            rb = batch.ResidentBatch(tb, derive_from=inp["lib"])  # --inputs host: hand-over (H2D + encoding) inside the timed region; resident: device-to-device
            t1 = time.perf_counter()
            res = rb.align(sensitivity=10, want_raw=True, n_threads=slot_threads)
            t2 = time.perf_counter()
            rb.close()
            with lock:
                agg["close_s"] = agg.get("close_s", 0.0) + (time.perf_counter() - t2)
                agg["lib_align_total_s"] = agg.get("lib_align_total_s", 0.0) + float(res.stats["total"])
            return res, t1 - t0, t2 - t1

        def on_result(ts, out, ta, tb_):
            res, c, a = out
            tr0 = time.perf_counter()
            with lock:
                agg["slot_s"] = agg.get("slot_s", 0.0) + (tb_ - ta)
                agg["create_s"] += c; agg["align_s"] += a
                st = res.stats
                agg["n_matches"] += int(st["n_matches"])
                add_stats(agg, st)
                agg["batches"].append((ta, tb_, len(ts), sum(t.bases for t in ts), int(st["n_matches"])))
                m = np.array(res.raw_matches, copy=True).view(MATCH_DTYPE)      # D2H of the matches is the library's; this is the host's copy of the list
                cg = np.array(res.raw_cigars, copy=True).view(np.uint32)
                if world == 1:
                    agg["results"].append((ts, m, cg, None))       # what the parity check digests after the timed region
                else:                                              # records keep the GLOBAL task id as their group
                    if len(m):
                        m["group"] = np.asarray([t.tid for t in ts], dtype=np.int32)[m["group"]]
                        m["cigar_off"] += np.uint64(pool_len[0])
                    recs.append(m); pools.append(cg); pool_len[0] += len(cg)
                agg["on_result_s"] = agg.get("on_result_s", 0.0) + (time.perf_counter() - tr0)
            res.close()

        if world == 1:
            sched.run_ready_set(tasks, run_batch, slots=args.slots, cap_bases=args.cap_gbp * 1e9, on_result=on_result, express=args.express, native=native_sched,
                                express_eps=float(os.environ.get("PGA_BENCH_EXPRESS_EPS", 0.05)), express_cap=float(os.environ.get("PGA_BENCH_EXPRESS_CAP", 60e6)))
            return agg
        # phase 1: this rank's subtrees, no communication; one gather.  Phase 2: the merges above the cut -- few, large, one after the other
        # along the tree -- by ALL ranks together: every rank indexes the whole call and maps its share of the queries of every group
        # (pga_batch_align_shard), level by level in an order every rank computes for itself; one more gather.
        tph0 = time.perf_counter()
        mine_ids = {t.tid for t in tasks if owner[t.tid] == rank}
        if mine_ids:
            sched.run_ready_set(tasks, run_batch, slots=args.slots, cap_bases=args.cap_gbp * 1e9, only=mine_ids, on_result=on_result, native=native_sched)

        def gather_all():
            t0 = time.perf_counter()
            m = np.concatenate(recs) if recs else np.zeros(0, MATCH_DTYPE)
            cg = np.concatenate(pools) if pools else np.zeros(0, np.uint32)
            pm = gather_blobs(m.view(np.uint8), cdev, dst=0, as_bytes=False)
            pc = gather_blobs(cg.view(np.uint8), cdev, dst=0, as_bytes=False)
            recs.clear(); pools.clear(); pool_len[0] = 0
            agg["gather_s"] += time.perf_counter() - t0
            if pm is None:
                return 0
            ident = np.arange(len(tasks), dtype=np.int32)
            rec, pool = merge_match_lists([t.numpy() for t in pm], [t.numpy() for t in pc], [ident] * len(pm))
            agg["results"].append((tasks, rec, pool, covered))   # rank 0: the gathered list, group = global task id
            return len(rec)

        covered = [t.tid for t in tasks if owner[t.tid] != -1]
        tph1 = time.perf_counter()
        n_gathered = gather_all()
        tph2 = time.perf_counter()
        top = [t.tid for t in tasks if owner[t.tid] == -1]
        done = {t.tid for t in tasks if owner[t.tid] != -1}
        while top:
            level = [tid for tid in top if all(d in done for d in tasks[tid].deps)]
            ts = [tasks[i] for i in sorted(level)]
            tb = sched.TaskBatch(ts, inp["first"])
            t0 = time.perf_counter()
            rb = batch.ResidentBatch(tb, derive_from=inp["lib"])
            t1 = time.perf_counter()
            res = rb.align(sensitivity=10, want_raw=True, n_threads=n_threads, shard=(rank, world))
            t2 = time.perf_counter()
            rb.close()
            on_result(ts, (res, t1 - t0, t2 - t1), t0, t2)
            done.update(level)
            top = [tid for tid in top if tid not in done]
        covered = [t.tid for t in tasks if owner[t.tid] == -1]
        tph3 = time.perf_counter()
        n_gathered += gather_all()
        agg["n_matches"] = n_gathered
        # this rank's phases (the gathers end when the slowest rank arrives: phase 1 of the slowest rank = phase1_s + gather1_s of the fastest)
        agg["phases"] = {"phase1_s": tph1 - tph0, "gather1_s": tph2 - tph1, "phase2_s": tph3 - tph2, "gather2_s": time.perf_counter() - tph3}
        return agg

    def step_waves():
        agg = {"create_s": 0.0, "align_s": 0.0, "gather_s": 0.0, "n_matches": 0, "stats": None, "per_wave": []}
        for w, pb in enumerate(mine):
            t0 = time.perf_counter()
            res = None
            if pb is not None:
                rb = batch.ResidentBatch(pb)                      # hand-over: inside the timed region
                t1 = time.perf_counter()
                res = rb.align(sensitivity=10, want_raw=world > 1, n_threads=n_threads)
                t2 = time.perf_counter()
                rb.close()
            else:
                t1 = t2 = t0
            if world > 1:
                z = np.zeros(0, np.uint8)
                got = gather_matches(res.raw_matches if res is not None else z, res.raw_cigars if res is not None else z, plans[w][rank], plans[w], cdev, dst=0)
                if got is not None:
                    agg["n_matches"] += len(got[0])
            t3 = time.perf_counter()
            agg["create_s"] += t1 - t0; agg["align_s"] += t2 - t1; agg["gather_s"] += t3 - t2
            if res is not None:
                st = res.stats
                if world == 1:
                    agg["n_matches"] += int(st["n_matches"])
                agg["per_wave"].append((waves[w][0], pb.total_bases, t1 - t0, t2 - t1, st["n_matches"]))
                add_stats(agg, st)
                res.close()
        return agg

    step = step_ready if args.schedule == "ready" else step_waves

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    batch.busy_begin()                                         # union of the launch intervals per kernel family over the timed steps
    cpu0 = os.times()
    t0 = time.perf_counter()
    last = None
    step_ends = []
    for _ in range(args.steps):
        last = step()
        step_ends.append(time.perf_counter())                  # (a step returns when its last batch has been collected: no extra synchronisation)
    torch.cuda.synchronize()
    busy = batch.busy_end()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    cpu1 = os.times()
    host_cpu_s = (cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)
    if world > 1:
        dt = max_over_ranks(dt, cdev)

    # ---- parity of the step that was timed: every find_matches call of the LAST timed step against the digests the compiled reference
    # produced for this build (tests/golden/builds_expected.json.gz).  A differing call means the number is not a measurement of the path.
    parity = {"parity_checked_calls": 0, "parity": "not checked"}
    build_sha = None
    if rank == 0 and args.schedule == "ready" and not args.leaf_only and last.get("results"):
        # one digest of the whole build's record lists (sha256 over the per-call digests in call order): equal for any number of ranks
        import hashlib
        from pangraph_amd import digest as dg
        per_call = {}
        for ts, rec, pool, covered in last["results"]:
            lists = dg.records_to_lists(rec, pool, [t.names for t in ts])
            for i in (range(len(ts)) if covered is None else covered):
                per_call[ts[i].tid] = (len(lists[i]), dg.digest(lists[i]))
        build_sha = hashlib.sha256(json.dumps([per_call[k] for k in sorted(per_call)]).encode()).hexdigest() if len(per_call) == len(tasks) else None
    if rank == 0 and args.schedule == "ready" and not args.no_parity_check and not args.leaf_only:
        from pangraph_amd import digest as dg
        gold = dg.expected_build(args.seed, args.genomes, args.length)
        if gold is None:
            parity["parity"] = "no golden build for these parameters (tests/golden/builds_expected.json.gz holds 1000 x 5 Mbp and 16 x 5.3 Mbp)"
        else:
            t_par = time.time()
            n_checked, bad = dg.check_calls(last["results"], dg.expected_by_call(pop, gold))
            parity = {"parity_checked_calls": n_checked, "parity": "every call of the last timed step == reference digest" if not bad else f"{len(bad)} calls differ",
                      "parity_check_s": round(time.time() - t_par, 2)}
            if bad or n_checked != len(tasks):
                print(json.dumps({"error": "parity check of the timed step failed: no value is reported", "calls_checked": n_checked, "calls_expected": len(tasks),
                                  "differing_calls_node_round": bad[:20]}))
                raise SystemExit(3)
    last["results"] = None

    # ---- secondary: the same step with the inputs already resident in HBM (N = 1, one step, outside the timed region)
    resident = None
    mem_after_timed = mem_stats()
    if want_secondary:
        inp.update(stash)
        step()                                                   # first use of the derive path: pools grow
        torch.cuda.synchronize()
        tr = time.perf_counter()
        step(); tr1 = time.perf_counter(); step()
        torch.cuda.synchronize()
        tr2 = time.perf_counter()
        res_steps = [1e3 * (tr1 - tr), 1e3 * (tr2 - tr1)]
        tr = (tr2 - tr) / 2
        resident = {"gbp_s": units / tr / 1e9, "ms_per_step": tr * 1e3, "steps": 2, "ms_of_each_step": res_steps, "inputs_made_resident_s": inp["t_lib"],
                    "note": "all sequences of every call in HBM (packed, 0.375 B/base) before the step; a call takes them device-to-device (pga_batch_derive)"}
        inp["lib"].close()
        inp["lib"], inp["first"] = None, None

    st = last["stats"]
    K = batch.KERNELS
    table = {}
    for i, name in enumerate(K):
        if name == "-" or st["kern_launches"][i] <= 0:
            continue
        ms, n, by, cells = st["kern_ms"][i], st["kern_launches"][i], st["kern_alg_bytes"][i], st["kern_cells"][i]
        e = {"device_ms_per_step": ms, "launches_per_step": n, "avg_launch_ms": ms / n,
             "busy_ms_per_step": busy.get(name, 0.0) / args.steps}      # union of its launch intervals over all streams and batches (rank 0)
        if batch.KERNEL_BOUND[i] == "hbm":
            gbs = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            e.update({"bound": "hbm", "alg_bytes_per_launch": by / n, "achieved_GBs": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS})
        else:
            e.update({"bound": "valu+lds (integer DP; no MFMA)", "alg_bytes_per_launch": by / n, "cells_evaluated": cells,
                      "gcups": cells / (ms * 1e-3) / 1e9 if ms > 0 else 0.0})
        table[name] = e
    kname = max(table, key=lambda k: table[k]["device_ms_per_step"])
    ki = K.index(kname)
    kms, klaunch, kbytes = st["kern_ms"][ki], st["kern_launches"][ki], st["kern_alg_bytes"][ki]
    achieved = (kbytes / klaunch) / (kms / klaunch * 1e-3) / 1e9 if klaunch and kms > 0 else 0.0
    dp_cells = sum(st["kern_cells"])
    dp_ms = sum(st["kern_ms"][i] for i in range(len(K)) if batch.KERNEL_BOUND[i] == "dp")
    ms_step = dt / args.steps * 1e3
    n_calls = len(tasks)
    timed = "pga_batch_create (H2D of the sequences + encoding) + pga_batch_align (sketch..records, D2H of matches)" if args.inputs == "host" or args.schedule != "ready" \
        else "pga_batch_derive (device-to-device) + pga_batch_align"
    # whole path: algorithmic HBM bytes per step by SURVEY 8d's formula with the measured counts
    alg_step = 1.5 * st["n_bases"] + 64.0 * st["n_minimizers"] + 52.0 * st["n_anchors"] + 0.5 * st["n_dp_bases"]
    valu_frac = None
    try:
        import glob
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_issue*kernels.json")))
        if f and kname.startswith("k_"):
            cand = [(v.get("SQ_WAVE_CYCLES", 0.0), v.get("valu_issue_frac_of_wave_cycles")) for k, v in json.load(open(f[-1])).get("kernels", {}).items()
                    if k.replace("pga::", "").split("<")[0] == kname.split("+")[0]]
            if cand:
                valu_frac = round(max(cand)[1], 3)       # the instantiation with the most wave cycles
    except Exception:   # noqa: BLE001
        valu_frac = None
    out = {
        "metric": "aligned Gbp/s in `pangraph build` (bases handed to the aligner per second, all merges, all self-merge rounds)",
        "value": units * args.steps / dt / 1e9,
        "unit": "Gbp/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{args.genomes} x {args.length} bp genomes, whole guide-tree build: {n_calls} find_matches calls "
                               f"({len(waves)} waves), U = {units / 1e9:.2f} Gbp per step (asm10 -c -X -s 90)" + (" [LEAF LEVEL ONLY]" if args.leaf_only else ""),
                   "genomes": args.genomes, "genome_length": args.length, "seed": args.seed,
                   "inputs": args.inputs, "timed_region": f"per batch: {timed}; + match-list gather (N > 1)",
                   "schedule": (f"ready set, {args.slots} batches in flight, decisions by " + ("the library (pga_sched_*)" if os.environ.get("PGA_NATIVE_SCHED", "1") not in ("", "0") else "schedule.ReadySet"))
                               if args.schedule == "ready" else "level-synchronous waves",
                   "parallelism": f"{world} rank(s): subtrees per rank, no data-path collective, match-list gathers only"},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "hbm_achieved_GBs": achieved, "hbm_frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(kname, klaunch),
                     "alg_bytes_per_launch": kbytes / klaunch if klaunch else None, "avg_launch_ms": kms / klaunch if klaunch else None,
                     "launches_per_step": klaunch, "busy_ms_per_step": busy.get(kname, 0.0) / args.steps,
                     "any_kernel_busy_ms_per_step": busy.get("any", 0.0) / args.steps,
                     "whole_path": {"alg_bytes_per_step": alg_step, "achieved": alg_step / (ms_step * 1e-3) / 1e9, "frac": alg_step / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                     "valu_issue_frac": valu_frac,
                     "note": "kernel = family with the largest summed HIP-event time; bound valu: VALU wave-instructions/s (SQ_INSTS_VALU of the committed PMC pass over this run's launch "
                             "time) against 1024 SIMD-32 x 2.4 GHz / 2; valu_model_*: round 5's self-referential cell model; rocprof_top_kernel, traffic, dispatches, waves: from profiles/, "
                             "not this run (DESIGN.md section 7)"},
        "n_matches_gathered": last["n_matches"],
        "build_sha256": build_sha,
        "resident_gbp_s": resident["gbp_s"] if resident else (units * args.steps / dt / 1e9 if inp["lib"] is not None else None),
    }
    # An integer DP kernel is bound by the VALU issue of its waves, not by HBM or MFMA.  frac is held against a HARDWARE peak: VALU wave-instructions per second
    # of the family (SQ_INSTS_VALU of the committed PMC pass of this workload, per step) over the family's summed launch time of THIS run, against what the chip
    # can issue (1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction).  The round-5 figure -- cells per second against a ceiling derived from the kernel's
    # own measured cycles per cell -- is self-referential (it says how much of its own arithmetic the kernel keeps busy, not how good that arithmetic is) and
    # stays beside it as valu_model_*; the HBM figures of the same kernel stay too.
    if batch.KERNEL_BOUND[ki] == "dp" and kms > 0:
        kcells = st["kern_cells"][ki]
        cyc = DP_CELL_CYCLES.get(kname.split("+")[0].split("<")[0], DP_CELL_CYCLES["k_extd2_lanes"])
        peak_cells = N_SIMD * CLOCK_HZ / cyc / 1e9
        ach_cells = kcells / (kms * 1e-3) / 1e9
        out["roofline"].update({"gcells_per_s": ach_cells, "cells_per_launch": kcells / klaunch if klaunch else None,
                                "valu_model_peak_gcells_per_s": peak_cells, "valu_model_frac": ach_cells / peak_cells, "cycles_per_cell_of_one_wave": cyc})
        vi = pmc_valu_insts([x for x in kname.replace(" ", "").split("+") if x.startswith("k_")])
        if vi:
            ach = vi / (kms * 1e-3)                      # wave-instructions per second while the family's launches run (summed launch time)
            out["roofline"].update({"bound": "valu", "achieved": ach / 1e9, "peak": VALU_PEAK_WAVE_INST_S / 1e9, "unit": "G wave-instructions/s", "frac": ach / VALU_PEAK_WAVE_INST_S,
                                    "valu_insts_per_cell": vi / kcells if kcells else None})
        # what a DP kernel moves through HBM is its direction matrix (a byte or two per cell, written for the backtrack and read by it), not the bases its
        # alg_bytes count: traffic per CELL says whether that is all it moves
        tr, cpl = out["roofline"].get("traffic"), out["roofline"].get("cells_per_launch")
        if tr and cpl:
            out["roofline"]["traffic_bytes_per_cell"] = tr / cpl
    out["roofline"]["rocprof_top_kernel"] = rocprof_top_kernel()
    out["roofline"].update(profile_counts(ms_step))
    out.update(parity)
    detail = {
        "bench_line": None,
        "rank0_seconds_per_step": {"hand_over": last["create_s"], "align": last["align_s"], "gather": last["gather_s"],
                                   "python_before_create": last.get("py_s"), "batch_close": last.get("close_s"), "on_result": last.get("on_result_s"), "slot_time_run_batch": last.get("slot_s"),
                                   "library_stage_total": last.get("lib_align_total_s"),
                                   "note": "summed over the batches in flight at the same time (ready-set schedule): not a decomposition of ms_per_step"},
        "roofline_note": "device_ms_per_step are HIP-event times on each kernel's own stream, summed over all launches; streams and batches overlap, so they do "
                         "not add up to ms_per_step -- busy_ms_per_step is the UNION of a family's launch intervals on the device clock (what the step spent with "
                         "that family queued or running), any_kernel_busy_ms_per_step the union over all families",
        "timed_intervals_per_step": busy.get("intervals", 0) / args.steps,
        "device_memory_cache_rank0": {"after_timed_steps": mem_after_timed, "at_exit": mem_stats()},
        "ms_of_each_timed_step_rank0": [round(1e3 * (b - a), 1) for a, b in zip([t0] + step_ends[:-1], step_ends)],
        "kernels": table,
        "dp": {"cells_evaluated": dp_cells, "gcups_over_dp_kernel_time": dp_cells / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0,
               "gcups_over_step": dp_cells / (ms_step * 1e-3) / 1e9, "nominal_cells_qlen_x_tlen": st["n_dp_cells"], "jobs": st["n_dp_jobs"],
               "issue_counters": pmc_issue()},
        "stages_s": {k: st[k] for k in ("upload", "sketch", "index", "seed", "chain", "align", "total")},
        "counts": {k: st[k] for k in ("n_bases", "n_minimizers", "n_anchors", "n_dp_jobs", "n_matches")},
        "aligned_span_gbp_s_rank0": st["aligned_span"] * args.steps / dt / 1e9,      # secondary: sum of (qe - qs) of rank 0's matches per second
        "waves_rank0": [{"wave": w, "Mbp": b / 1e6, "hand_over_s": round(c, 4), "align_s": round(a, 4), "matches": int(m)} for w, b, c, a, m in last["per_wave"]],
        "batches_rank0": [{"t0": round(a, 4), "t1": round(b, 4), "calls": n, "Mbp": round(bs / 1e6, 1), "matches": m} for a, b, n, bs, m in sorted(last.get("batches", []))],
        "workload_generation_s": t_gen,
        "host_cpu": {"cpu_s_per_step": host_cpu_s / args.steps, "system_s_per_step": (cpu1.system - cpu0.system) / args.steps, "mean_busy_cores": host_cpu_s / dt, "usable_cores": usable_cpus(), "threads_per_batch": slot_threads},
        "resident_inputs": resident,
        "predicted_scaling": (sched.predict_scaling(pop, tasks, (1, 2, 4, 8), units * args.steps / dt / 1e9, args.slots, host_cpu_s_per_gbp=host_cpu_s / args.steps / (units / 1e9),
                                                    host_cores=usable_cpus()) if world == 1 and args.schedule == "ready" and not args.leaf_only else None),
        "traffic_source": "profiles/ (rocprofv3 --pmc passes of this workload, not this run); bytes per launch in the unit of alg_bytes_per_launch",
    }
    if world > 1 and args.schedule == "ready" and last.get("phases"):
        # the phases of the last timed step on rank 0, beside what schedule.predict_scaling says for this world size from a single-GPU rate
        # (--model-gbp-s / --model-host-s-per-gbp, defaults: the committed single-GPU line).  With PGA_BENCH_SINGLE_DEVICE=1 all ranks share ONE device:
        # the measured times then say how the phases relate, not how fast N devices are.
        model = sched.predict_scaling(pop, tasks, (world,), args.model_gbp_s, args.slots, host_cpu_s_per_gbp=args.model_host_s_per_gbp, host_cores=usable_cpus())[str(world)]
        detail["phases_rank0"] = {"measured_s": {k: round(v, 3) for k, v in last["phases"].items()},
                                  "model_s": {k: model[k] for k in ("phase1_s", "phase2_s", "step_s", "phase1_bound", "host_cores_per_rank", "calls_above_the_cut")},
                                  "model_inputs": {"gbp_s_one_gpu": args.model_gbp_s, "host_cpu_s_per_gbp": args.model_host_s_per_gbp, "host_cores": usable_cpus()},
                                  "single_device": bool(os.environ.get("PGA_BENCH_SINGLE_DEVICE"))}
    if rank == 0:
        if args.cpu_budget > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(waves, args.cpu_budget)
            detail["cpu_baseline_one_core_gbp_s"] = out["cpu_baseline"].pop("one_core_gbp_s", None)
        elif world > 1:
            out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": 0, "kind": "reference", "sample": "measured at N=1 only"}
        if world == 1 and not args.no_next_rows:
            detail["next_rows"] = next_rows(pop, args.seed)
        line = json.dumps(out)
        if len(line) > 4000:                                       # the driver parses ONE short line: drop the prose first
            out["roofline"].pop("note", None)
            out["config"].pop("parallelism", None)
            line = json.dumps(out)
        detail["bench_line"] = out
        try:
            with open(args.detail, "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as e:
            print(f"bench.py: detail not written: {e}", file=sys.stderr)
        print(line, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
