#!/bin/bash
# dev tool (GPU box): FETCH_SIZE / WRITE_SIZE of the merge kernel for library variants.  usage: bash tools/pmc_fetch_ab.sh [-a "<kbench args>"] a.so b.so ...
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo"
extra=""
if [ "$1" = "-a" ]; then extra=$2; shift 2; fi
for lib in "$@"; do
  tag=$(basename $lib .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pf_${tag}_$c
    KMD_LIB=$repo/$lib timeout 300 rocprofv3 --pmc $c -d gpurun_out/pf_${tag}_$c -o pmc --output-format csv -- python3 tools/kbench_pipeline.py --fused-only --iters 2 $extra > gpurun_out/pf_$tag.log 2>&1 < /dev/null
  done
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
out = []
for c, corr in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    acc, n = 0.0, 0
    for f in glob.glob('gpurun_out/pf_%s_%s/**/*counter_collection.csv' % (tag, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_tile_sums<512' in r['Kernel_Name'] and r['Counter_Name'] == c:
                acc += float(r['Counter_Value']); n += 1
    out.append("%s %.4e B/launch (%d launches; KB x1024 x%.0f)" % (c, acc / max(n, 1) * 1024 * corr, n, corr))
print("%-10s %s" % (tag, "; ".join(out)))
PY
done
