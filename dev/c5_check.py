"""Full-size validation of the BASELINE workload (config C5): waves of Population(20260928, N, 5 Mbp) through the native batch
entry vs the reference build, group by group.  usage: python dev/c5_check.py [n_genomes=1000] [wave indices, comma separated | all]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from pangraph_amd.levels import Population
from levels_util import ref_align_groups, product_align_groups
def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    t0 = time.time()
    waves = Population(20260928, n, 5_000_000).build_waves()
    print(f"generated {len(waves)} waves in {time.time()-t0:.0f} s", flush=True)
    sel = range(len(waves)) if len(sys.argv) < 3 or sys.argv[2] == "all" else [int(x) for x in sys.argv[2].split(",")]
    bad = 0
    for w in sel:
        label, groups, names = waves[w]
        got, want = [], []
        t_gpu = t_ref = 0.0
        for lo in range(0, len(groups), 48):          # slices bound the host memory of the reference's worker processes
            t0 = time.time(); got += product_align_groups(groups[lo:lo + 48], names[lo:lo + 48], sensitivity=10); t1 = time.time()
            want += ref_align_groups(groups[lo:lo + 48], names[lo:lo + 48], sensitivity=10); t2 = time.time()
            t_gpu += t1 - t0; t_ref += t2 - t1
        t0, t1, t2 = 0.0, t_gpu, t_gpu + t_ref
        nb = sum(1 for a, b in zip(got, want) if a != b)
        bad += nb
        print(f"wave {w} {label}: {sum(len(x) for x in want)} records, gpu {t1-t0:.1f} s, ref {t2-t1:.1f} s, {nb} of {len(groups)} groups differ", flush=True)
        for g, (a, b) in enumerate(zip(got, want)):
            if a != b:
                print("   group", g, len(a), "vs", len(b), [x for x in a if x not in b][:2], [x for x in b if x not in a][:2], flush=True)
    print("TOTAL groups differing:", bad)


if __name__ == '__main__':      # (the reference's worker processes are spawned: they re-import this module)
    main()
