cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfg in "4096 18" "8192 19" "16384 20" "32768 21"; do
  set -- $cfg
  rm -rf gpurun_out/co_$1
  KMD_TILE_COARSE=$1 KMD_TILE_COARSE_CELLS=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/co_$1 -o t -- python3 tools/kbench_pipeline.py --device --rows 39062500 --iters 6 > gpurun_out/co_$1.log 2>&1 < /dev/null
  call=$(grep -o "kmd_merge_filter) [0-9.]* ms" gpurun_out/co_$1.log | head -1)
  python3 - $1 "$call" <<'PY'
import csv, glob, sys
tag, call = sys.argv[1], sys.argv[2]
for f in glob.glob('gpurun_out/co_%s/**/*kernel_stats.csv' % tag, recursive=True):
    rows = list(csv.DictReader(open(f)))
    out = ['%s %.1fus' % (k, float(r['AverageNs']) / 1e3) for r in rows for k in ('k_tile_sums<512', 'k_tile_probe', 'k_tile_bounds', 'k_tile_index') if k in r['Name']]
    print(tag, call, '; '.join(out))
PY
done
