#!/usr/bin/env python3
"""bench.py -- aligned Gbp/s of the pairwise block-alignment backend on synthetic genomes (BASELINE.json metric).

A STEP is one pass of the whole hot path (sketch -> index -> seed/anchor -> chain -> banded DP extension ->
records) over one guide-tree LEVEL: every rank aligns its own G all-vs-all groups in ONE batch.  The workload is
the leaf level of `pangraph build` on synthetic bacterial-like genomes (SURVEY.md section 8d generator: random
ancestor, SNPs, indels, inversions, HGT-like insertions, deletions): group g = two sibling genomes, exactly the
block set of the first `find_matches` call of that merge (graph_merging.rs:98).  Units U = bases handed to the
aligner (sum of sequence lengths of all groups); value = U * steps / wall, summed over ranks
(weak scaling: per-rank work is fixed).  Sequences are copied to HBM before the timed region (pga_batch_create);
the timed region runs pga_batch_align() and, for N>1, the RCCL gather of the match lists to rank 0.

The line also carries
  roofline      the dominant kernel of the run (largest device time): algorithmic bytes / HIP-event time vs 8 TB/s
  cpu_baseline  the REFERENCE's own C (oracle/_ref/libmm2ref.so, compiled from /root/reference by oracle/Makefile),
                one process per host core, on a bounded sample of the same groups
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import pangraph_amd  # noqa: E402  (sets the HIP runtime defaults of the backend before anything initialises HIP)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_groups(seed: int, n_genomes: int, length: int, div: float):
    """n_genomes/2 sibling pairs; every pair descends from its own ancestor (leaf merges are independent)."""
    import numpy as np
    from pangraph_amd.synth import random_seq, mutate
    groups, names = [], []
    for g in range(n_genomes // 2):
        rng = np.random.default_rng(seed * 1000003 + g)
        anc = random_seq(rng, length)
        ev = max(2000, min(50000, length // 40))
        kids = [mutate(rng, anc, snp=div / 2, indel=div / 20, n_inv=2, n_ins=6, n_del=4, max_event=ev) for _ in range(2)]
        groups.append([k.tobytes() for k in kids])
        names.append([str(2 * g), str(2 * g + 1)])
    return groups, names


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


PMC_KERNEL_PREFIX = {"k_sketch_tiles": "void pga::k_sketch_tiles", "k_chain_fast": "void pga::k_chain_fast", "k_bt_list+k_bt_walk": "pga::k_bt_",
                     "k_extd2_fast": "void pga::k_extd2_fast", "k_extd2_wide": "void pga::k_extd2_wide", "k_ll_i16": "pga::k_ll_i16", "k_rs_pass": "pga::k_rs_",
                     "k_gapfill_band": "pga::k_gapfill_band"}


def pmc_traffic(kernel: str, genomes: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same workload
    (profiles/r01_*_pmc_hbm_traffic_<genomes>genomes.json: FETCH_SIZE and WRITE_SIZE in separate passes, in KB; FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for gfx950 -- an upper bound for the narrower accesses).  None if there is no
    PMC summary for this configuration."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_hbm_traffic_{genomes}genomes.json")))
    pre = PMC_KERNEL_PREFIX.get(kernel)
    if not files or not pre:
        return None
    d = json.load(open(files[-1]))
    tot, n = 0.0, 0
    for c, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        disp = 0
        for k, v in d.get(c, {}).items():
            if k.startswith(pre):
                tot += mul * v["sum"] * 1024.0
                disp += v["dispatches"]
        n = max(n, disp)
    return tot / n if n else None


def _cpu_worker(args):
    so, seqs, names = args
    from pangraph_amd.mm2ffi import Mm2Lib
    lib = Mm2Lib(so)
    t0 = time.time()
    rows = lib.align_all([s.decode() for s in seqs], names, sensitivity=10)
    return time.time() - t0, sum(len(s) for s in seqs), len(rows)


def cpu_baseline(groups, names, budget_s: float):
    """The reference C on the host cores: one group per process, as many groups as cores (bounded sample)."""
    import multiprocessing as mp
    so = os.path.join(ROOT, "oracle", "_ref", "libmm2ref.so")
    kind = "reference"
    if not os.path.exists(so):
        so, kind = os.path.join(ROOT, "oracle", "libpgoracle.so"), "port"
    cores = usable_cpus()
    n = min(cores, len(groups))
    # calibrate on one group, then size the sample to the budget
    t1, b1, _ = _cpu_worker((so, groups[0], names[0]))
    rounds = max(1, min(4, int(budget_s / max(t1, 1e-3))))
    jobs = [(so, groups[i % len(groups)], names[i % len(groups)]) for i in range(n * rounds)]
    t0 = time.time()
    with mp.get_context("spawn").Pool(n) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.time() - t0
    bases = sum(r[1] for r in res)
    return {"value": bases / wall / 1e9, "unit": "Gbp/s", "cores": n, "kind": kind,
            "sample": f"{len(jobs)} leaf-pair groups ({bases / 1e6:.1f} Mbp), {n} processes x 1 thread, {wall:.1f} s; 1 core: {b1 / t1 / 1e9:.5f} Gbp/s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=int(os.environ.get("PGA_BENCH_GENOMES", 512)), help="genomes PER GPU at the leaf level")
    ap.add_argument("--length", type=int, default=int(os.environ.get("PGA_BENCH_LENGTH", 5_000_000)))
    ap.add_argument("--divergence", type=float, default=0.01)
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU-baseline work (0 disables)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pangraph_amd import batch
    from pangraph_amd.dist import gather_blobs, max_over_ranks, sum_over_ranks

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    # PGA_BENCH_SINGLE_DEVICE=1 (debugging on a 1-GPU box): every rank computes on GPU 0 and the collectives run over gloo
    single = os.environ.get("PGA_BENCH_SINGLE_DEVICE") == "1"
    if single:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    batch.set_device(local)
    if world > 1:
        if single:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cdev = torch.device("cpu") if single else dev          # where the collectives' tensors live

    groups, names = make_groups(20260928 + rank, args.genomes, args.length, args.divergence)
    pb = batch.PreparedBatch(groups, names)
    rb = batch.ResidentBatch(pb)                     # H2D happens here, outside the timed region
    units = float(pb.total_bases)

    # host threads of this rank: the ranks of a node share one CPU quota
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    n_threads = max(2, usable_cpus() // max(1, local_world))

    def step(want_raw):
        res = rb.align(sensitivity=10, want_raw=want_raw, n_threads=n_threads)
        if world > 1:
            # the match list (records, then the CIGAR pool) goes to the rank that owns the graph
            gather_blobs(res.raw_matches, cdev, dst=0, as_bytes=False)
            gather_blobs(res.raw_cigars, cdev, dst=0, as_bytes=False)
        res.close()
        return res

    for _ in range(args.warmup):
        step(world > 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step(world > 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = max_over_ranks(dt, cdev)
        total_units = sum_over_ranks(units, cdev)
    else:
        total_units = units

    st = last.stats
    kern = [(st["kern_ms"][i], batch.KERNELS[i], st["kern_launches"][i], st["kern_alg_bytes"][i]) for i in range(len(batch.KERNELS)) if st["kern_launches"][i] > 0]
    kms, kname, klaunch, kbytes = max(kern)
    achieved = (kbytes / klaunch) / (kms / klaunch * 1e-3) / 1e9 if klaunch and kms > 0 else 0.0
    out = {
        "metric": "aligned Gbp/s in the pangraph-build alignment backend (bases handed to the aligner per second)",
        "value": total_units * args.steps / dt / 1e9,
        "unit": "Gbp/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"leaf level of pangraph build: {args.genomes // 2} sibling-genome pairs per GPU x {args.length} bp "
                               f"(asm10, -c -X -s 90, ~{args.divergence * 100:.1f}% divergence + inversions/HGT/deletions), one batch per step",
                   "genomes_per_gpu": args.genomes, "genome_length": args.length, "groups_per_gpu": args.genomes // 2,
                   "parallelism": f"groups sharded over {world} rank(s), match-list gather to rank 0"},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(kname, args.genomes), "launches_per_step": klaunch, "avg_launch_ms": kms / klaunch if klaunch else None,
                     "alg_bytes_per_launch": kbytes / klaunch if klaunch else None},
        "stages_s": {k: st[k] for k in ("sketch", "index", "seed", "chain", "align", "total")},
        "kernels_ms": {batch.KERNELS[i]: st["kern_ms"][i] for i in range(len(batch.KERNELS)) if st["kern_launches"][i] > 0},
        "counts": {k: st[k] for k in ("n_bases", "n_minimizers", "n_anchors", "n_dp_jobs", "n_dp_cells", "n_matches")},
        "aligned_span_gbp_s_per_gpu": st["aligned_span"] * args.steps / dt / 1e9,      # secondary: sum of (qe - qs) of this rank's matches per second
    }
    if rank == 0:
        if args.cpu_budget > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(groups, names, args.cpu_budget)
        elif world > 1:
            out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": 0, "kind": "reference", "sample": "measured at N=1 only"}
        print(json.dumps(out))
    rb.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
