#!/bin/bash
# On the GPU box: kernel trace and HBM traffic counters of kmd_merge_filter on one WHOLE configs[2] partition
# (39 062 500 rows, ~1e9 records; streams built on the device).  usage: bash tools/prof_c3.sh [out dir] [extra kbench args]
repo=${GRAFT_REPO_ROOT:-$PWD}
O=${1:-gpurun_out/r04/c3}; shift
cd /tmp && export TMPDIR=/tmp
cd "$repo" && mkdir -p $O && rm -rf $O/trace $O/pmc_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o pipe -- python3 tools/kbench_pipeline.py --device --rows 39062500 --iters 6 "$@" > $O/trace.log 2>&1 < /dev/null
tail -1 $O/trace.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 tools/kbench_pipeline.py --device --rows 39062500 --iters 2 "$@" > $O/pmc_$c.log 2>&1 < /dev/null
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:12]:
        print("%-100s calls %4s avg %10.1f us  total %6.2f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(O + "/pmc_" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    for k in sorted(acc, key=lambda k: -acc[k])[:4]:
        corr = 2.0 if c == "FETCH_SIZE" else 1.0
        print("%s %-90s %.1f KB/launch (%d launches) -> %.4e B (x1024 x%.0f)" % (c, k[:90], acc[k] / n[k], n[k], acc[k] / n[k] * 1024 * corr, corr))
PY
