"""The index stage with and without the candidate route (PGA_INDEX_BUCKETS=1, pga_index_buckets.h): device time of the index build (K_INDEX slot of
pga_stats_t: HIP events around build_index_ex) for a leaf-like batch (whole-genome pairs) and an upper-tree-like batch (many groups of blocks), the
records compared.  The switch is read once per process, so the script runs itself twice.
    gpurun -- 'python dev/index_probe.py'            (about a minute)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def child():
    from pangraph_amd import batch
    from pangraph_amd.levels import Population
    from pangraph_amd.digest import digest
    from util import rows_to_lists
    ki = batch.KERNELS.index("index build (sorts + CSR kernels)")
    out = {}
    pop = Population(20260928, 24, 5_000_000)
    waves = pop.build_waves()
    for tag, w in (("leaf level: 12 whole-genome pairs", 0), ("tree height 3, round 0", 4), ("root, round 0", len(waves) - 2)):
        label, groups, names = waves[w]
        best, dig = None, None
        for rep in range(3):
            res = batch.align_groups([[a.tobytes() for a in g] for g in groups], names, sensitivity=10, want_rows=rep == 0)
            st = res.stats
            ms = st["kern_ms"][ki]
            best = ms if best is None else min(best, ms)
            if rep == 0:
                dig = [digest(rows_to_lists(r)) for r in res.groups]
            out[tag] = {"wave": label, "groups": len(groups), "minimizers": st["n_minimizers"], "index_ms_best_of_3": best, "index_stage_s": st["index"], "digest": dig}
            res.close()
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    got = {}
    for tag, env in (("sort route", {}), ("bucket route", {"PGA_INDEX_BUCKETS": "1"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        if r.returncode != 0 or not line:
            print(tag, "FAILED rc", r.returncode, r.stderr[-2000:]); continue
        got[tag] = json.loads(line[-1][7:])
    for k in got.get("sort route", {}):
        a, b = got["sort route"][k], got.get("bucket route", {}).get(k)
        print(f"{k}: {a['groups']} groups, {a['minimizers']:.0f} minimizers | sort route {a['index_ms_best_of_3']:.3f} ms"
              + (f" | bucket route {b['index_ms_best_of_3']:.3f} ms | records {'identical' if a['digest'] == b['digest'] else 'DIFFER'}" if b else ""))
