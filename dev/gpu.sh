#!/bin/bash
# ONE launcher for everything that runs on the GPU box (replaces the per-experiment r0N_*.sh scripts of earlier rounds).
#   gpurun -- 'dev/gpu.sh tests [pytest args]'            the -m gpu suite
#   gpurun -- 'dev/gpu.sh bench <name> [bench args]'      default bench line -> gpurun_out/<name>.json (+ _detail.json, .err); env vars pass through
#   gpurun -- 'dev/gpu.sh ab <name> "ENV=.. ENV=.." ...'  A/B: one short bench per quoted environment
#   gpurun -- 'dev/gpu.sh profile <tag> [stats|traffic|issue ...]'   rocprofv3 summaries of the BASELINE workload -> gpurun_out/profiles_out/<ROUND>_<tag>_*
#   gpurun -- 'dev/gpu.sh verbose <name> [bench args]'    one step under PGA_VERBOSE=1, stderr gzipped
# ROUND (default r04) prefixes the profile files; copy gpurun_out/profiles_out/* into profiles/ to have them judged.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
R=$PWD
ROUND=${ROUND:-r06}
mkdir -p gpurun_out/profiles_out
cmd=$1; shift

summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    try:
        dd = json.load(open(sys.argv[1].replace(".json", "_detail.json"))); hc = dd["host_cpu"]; ms = sorted(dd.get("ms_of_each_timed_step_rank0", []))
        print("   host cpu: %.1f cores busy on average of %d (system %.1f s of %.1f s per step) | steps ms: min %.0f median %.0f max %.0f (n=%d)" % (hc["mean_busy_cores"], hc["usable_cores"], hc.get("system_s_per_step", -1), hc["cpu_s_per_step"], ms[0], ms[len(ms) // 2], ms[-1], len(ms)))
    except Exception: pass
    print(sys.argv[1], "value", round(d["value"], 3), "ms", round(d["ms_per_step"]), "resident", d.get("resident_gbp_s"), "parity", d.get("parity_checked_calls"),
          "| kernel", r.get("kernel"), "frac", r.get("frac"), "busy", r.get("busy_ms_per_step"), "any", r.get("any_kernel_busy_ms_per_step"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), "| line bytes", len(json.dumps(d)))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
}

case $cmd in
tests)
  timeout ${T:-2400} python -m pytest tests -m gpu -x -q --timeout 1500 "$@" 2>&1 | tail -${TAILN:-8} ;;
bench)
  name=$1; shift
  timeout ${T:-900} python bench.py --detail gpurun_out/${name}_detail.json "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err; echo "bench rc=$?"
  tail -3 gpurun_out/$name.err; summ gpurun_out/$name.json ;;
ab)
  name=$1; shift
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout ${T:-600} python bench.py --steps ${STEPS:-2} --warmup 1 --cpu-budget 0 --no-next-rows --no-resident-rate ${BENCH_ARGS:-} --detail gpurun_out/${name}_${i}_detail.json > gpurun_out/${name}_$i.json 2> gpurun_out/${name}_$i.err
    echo "[$i] $e rc=$?"; summ gpurun_out/${name}_$i.json
  done ;;
verbose)
  name=$1; shift
  PGA_VERBOSE=1 timeout ${T:-600} python bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-next-rows --no-resident-rate "$@" --detail gpurun_out/${name}_detail.json > gpurun_out/$name.json 2> gpurun_out/$name.err; echo "rc=$?"
  gzip -f gpurun_out/$name.err; summ gpurun_out/$name.json ;;
profile)
  TAG=$1; shift
  what=${*:-stats traffic issue}
  BA="--cpu-budget 0 --no-next-rows --no-resident-rate --no-parity-check"
  for w in $what; do case $w in
  stats)
    ( cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python $R/bench.py $BA --steps 1 --warmup 1 --detail $R/gpurun_out/profiles_out/${ROUND}_${TAG}_bench_under_rocprof_detail.json > $R/gpurun_out/profiles_out/${ROUND}_${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_$TAG.err ); echo "stats rc=$?"
    f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/profiles_out/${ROUND}_${TAG}_c5_kernel_stats.csv
    kt=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
    [ -n "$kt" ] && python dev/concurrency.py "$kt" gpurun_out/profiles_out/${ROUND}_${TAG}_kernel_concurrency.json
    find gpurun_out/prof_$TAG \( -name "*.db" -o -name "*kernel_trace.csv" \) -size +20M -delete
    head -22 gpurun_out/profiles_out/${ROUND}_${TAG}_c5_kernel_stats.csv | cut -c1-160 ;;
  traffic)
    for c in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && timeout 1500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/bench.py $BA --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_$c.err ); echo "$c rc=$?"
    done
    python - <<PY
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print("no counter file for", c); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0]
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    out[c] = {k: {"sum": v[0], "dispatches": v[1]} for k, v in agg.items()}
    top = sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]
    print(c, "total KB %.4g" % sum(v[0] for v in agg.values())); [print("  %-70s sum %.4g KB over %d dispatches" % (k[:70], v[0], v[1])) for k, v in top]
out["_meta"] = {"steps": 1, "warmup": 0, "note": "one step of the BASELINE build per pass (bench.py defaults: --inputs host, ready set, six slots); sums are KB over all dispatches of the step; FETCH_SIZE is doubled by the reader (MI355X_MICROARCH.md, gfx950)"}
json.dump(out, open("gpurun_out/profiles_out/${ROUND}_${TAG}_pmc_hbm_traffic_c5.json", "w"), indent=1)
PY
    find gpurun_out/pmc_* -name "*.csv" -size +20M -delete ;;
  issue)
    for c in SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU; do
      ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmci_$c -o pmc -- python $R/bench.py $BA --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmci_$c.err ); echo "$c rc=$?"
    done
    python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(dict)
for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU"):
    f = glob.glob(f"gpurun_out/pmci_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print("no counter file for", c); continue
    tot = collections.defaultdict(float)
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != c: continue
            tot[row["Kernel_Name"].split("(")[0].replace("void ", "")] += float(row["Counter_Value"])
    for k, v in tot.items(): agg[k][c] = v
out = {}
all_wc = sum(v.get("SQ_WAVE_CYCLES", 0.0) for v in agg.values())
for k, v in agg.items():
    if "k_" not in k: continue
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    v["valu_issue_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc if wc else None
    v["lds_issue_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_LDS", 0.0) / wc if wc else None
    out[k] = v
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:24]:
    print("%-44s VALU issue %.3f  LDS issue %.3f  of wave cycles;  insts VALU %.3g SALU %.3g" % (k[:44], v["valu_issue_frac_of_wave_cycles"] or 0, v["lds_issue_frac_of_wave_cycles"] or 0, v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0)))
print("SQ_WAVE_CYCLES of all kernels of the step (quad-cycles): %.4g" % all_wc)
json.dump({"_source": "dev/gpu.sh profile <tag> issue: one step of the BASELINE build per counter pass (six batches in flight)", "all_kernels_SQ_WAVE_CYCLES": all_wc, "kernels": out}, open("gpurun_out/profiles_out/${ROUND}_${TAG}_pmc_issue_kernels.json", "w"), indent=1)
PY
    find gpurun_out/pmci_* -name "*.csv" -size +20M -delete ;;
  esac; done ;;
*) echo "usage: dev/gpu.sh tests|bench|ab|verbose|profile ..."; exit 2 ;;
esac
