#!/bin/bash
# dev tool (GPU box): kernel timeline of one kmd_merge_filter call (rocprofv3 kernel trace + memory copies are not traced: gaps show them)
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo" && rm -rf gpurun_out/prof_tile && mkdir -p gpurun_out
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tile -o t -- python3 tools/kbench_pipeline.py --fused-only --iters 8 "$@" \
  > gpurun_out/prof_tile.log 2>&1 < /dev/null
grep fused gpurun_out/prof_tile.log
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_tile/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_tile_probe" in r["Kernel_Name"]]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev = None
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%6.1f gap  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3 if prev else 0, (e - s) / 1e3, r["Kernel_Name"][:70]))
    prev = e
PY
