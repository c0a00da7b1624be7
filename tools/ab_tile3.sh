#!/bin/bash
# dev tool (GPU box): kernel-trace A/B of library variants on the fused merge + test (one 20v20 partition, 104 M records).
# usage: bash tools/ab_tile3.sh [-a "<kbench_pipeline args>"] build_sweep/a.so build_sweep/b.so ...   -> one line per variant
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo"
extra=""
if [ "$1" = "-a" ]; then extra=$2; shift 2; fi
for lib in "$@"; do
  tag=$(basename $lib .so)
  rm -rf gpurun_out/ab_$tag
  KMD_LIB=$repo/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$tag -o t -- python3 tools/kbench_pipeline.py --fused-only --iters 8 $extra \
    > gpurun_out/ab_$tag.log 2>&1 < /dev/null
  call=$(grep -o "kmd_merge_filter) [0-9.]* ms" gpurun_out/ab_$tag.log | head -1)
  python3 - "$tag" "$call" <<'PY'
import csv, glob, sys
tag, call = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob('gpurun_out/ab_%s/**/*kernel_stats.csv' % tag, recursive=True):
    rows = list(csv.DictReader(open(f)))
    break
out = []
for r in rows:
    n = r['Name']
    for key in ('k_tile_sums', 'k_tile_probe', 'k_tile_bounds', 'k_cand_eval', 'k_cand_scan', 'k_cand_emit', 'k_resolve_near', 'k_tile_index', 'k_tile_refine'):
        if key in n:
            short = key + ('<' + n.split('<')[1].split('>')[0] + '>' if key == 'k_tile_sums' else '')
            out.append('%s %sx %.1fus' % (short, r['Calls'], float(r['AverageNs']) / 1e3))
print('%-10s %s | %s' % (tag, call, '; '.join(out)))
PY
done
