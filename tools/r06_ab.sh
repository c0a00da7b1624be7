#!/bin/bash
# dev (GPU box): kernel-trace A/B of library variants on the fused merge (kmd_merge_filter), the shapes VERDICT r5 names.
# usage: tools/r06_ab.sh [-s "1 2 3 4 5 6"] build_sweep/a.so ...      (summary appended to gpurun_out/r06_ab/summary.txt)
#   1: rows of 2.9 records (36 M rows)   2: rows of 7.8 records (13 M rows)   3: 4 M-row 20v20   4: one configs[2] partition
#   5: 4v4 10^8 rows (configs[1])        6: the MIXED configs[2]-size partition
shapes="1 2 3 4 5"
if [ "$1" = "-s" ]; then shapes=$2; shift 2; fi
out=gpurun_out/r06_ab; mkdir -p $out
A1="--fused-only --sparse 0.1 --rows 40000000 --iters 3"
A2="--fused-only --sparse 0.3 --rows 13333333 --iters 3"
A3="--fused-only --iters 6"
A4="--device --rows 39062500 --iters 5"
A5="--device --rows 100000000 --nc 4 --nk 4 --iters 5"
A6="--device --rows 39062500 --iters 5 --profile 1"
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp; cd "$repo"
for shape in $shapes; do
  eval args=\$A$shape
  echo "== shape $shape: $args" | tee -a $out/summary.txt
  for lib in "$@"; do
    tag=$(basename $lib .so)
    rm -rf $out/t_${tag}_$shape
    KMD_LIB=$repo/$lib timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_${tag}_$shape -o t -- python3 tools/kbench_pipeline.py $args > $out/${tag}_$shape.log 2>&1 < /dev/null
    call=$(grep -o "kmd_merge_filter) [0-9.]* ms.*GB/s of [0-9]* B/record" $out/${tag}_$shape.log | head -1 | sed -e 's/ [0-9.e+]* rows\/s  [0-9.e+]* records\/s//')
    python3 - "$tag" "$call" "$out/t_${tag}_$shape" <<'PY' | tee -a $out/summary.txt
import csv, glob, sys
tag, call, d = sys.argv[1], sys.argv[2], sys.argv[3]
rows = []
for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    break
out = []
for r in rows:
    n = r['Name']
    for key in ('k_tile_sums', 'k_tile_probe', 'k_tile_bounds', 'k_tile_index', 'k_cand_eval', 'k_cand_emit', 'k_cand_scan', 'k_resolve_near'):
        if key in n and float(r['AverageNs']) > 4000:
            short = key + ('<' + n.split('<')[1].split('>')[0].replace(' ', '') + '>' if key == 'k_tile_sums' else '')
            out.append('%s %sx %.1fus' % (short, r['Calls'], float(r['AverageNs']) / 1e3))
print('%-12s %s | %s' % (tag, call, '; '.join(out)))
PY
  done
done
