#!/bin/bash
# rocprofv3 kernel trace of the fused merge + test on one 20v20 partition -> gpurun_out/prof_tile/*kernel_stats.csv
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo" && rm -rf gpurun_out/prof_tile && mkdir -p gpurun_out
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tile -o t -- python3 tools/kbench_pipeline.py --fused-only --iters 8 "$@" \
  > gpurun_out/prof_tile.log 2>&1 < /dev/null
grep fused gpurun_out/prof_tile.log
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_tile/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.reader(open(f)))[:9]:
        print(r[0][:70], r[1:5])
    break
PY
