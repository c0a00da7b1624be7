#!/bin/bash
# GPU box: kernel trace of six partitions in flight (tools/kbench_pipeline.py --overlap 6): how much of the merge
# kernels' time other kernels of other streams run beside them -> gpurun_out/r02/overlap.txt (+ kernel stats CSV)
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo" && mkdir -p gpurun_out/r02 && rm -rf gpurun_out/r02/prof_overlap
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof_overlap -o ovl -- python3 tools/kbench_pipeline.py --fused-only --iters 6 --overlap 6 \
  > gpurun_out/r02/prof_overlap.log 2>&1 < /dev/null
python3 - <<'PY' > gpurun_out/r02/overlap.txt
import csv, glob
f = glob.glob('gpurun_out/r02/prof_overlap/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
# the timed part: the last 36 launches of the wide merge kernel (6 streams x 6 calls)
merges = [e for e in ev if "k_tile_sums" in e[2] and "true, false, true" in e[2]]
last = merges[-36:]
t0, t1 = last[0][0], max(e[1] for e in last)
inside = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy = sum(e[1] - e[0] for e in inside)
merge_busy = sum(e[1] - e[0] for e in last)
# wall time covered by at least one kernel
cov, cur_s, cur_e = 0, None, None
for s, e, _, _ in inside:
    if cur_e is None or s > cur_e:
        if cur_e is not None: cov += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None: cov += cur_e - cur_s
print("six partitions in flight, 36 calls: wall %.3f ms = %.3f ms per partition" % ((t1 - t0) / 1e6, (t1 - t0) / 36e6))
print("kernel time summed over all streams %.3f ms (merge kernels %.3f ms = %.3f ms each), wall covered by at least one kernel %.3f ms"
      % (busy / 1e6, merge_busy / 1e6, merge_busy / 36e6, cov / 1e6))
print("=> %.0f %% of the kernel time ran beside another kernel; streams seen: %d" % (100.0 * (busy - cov) / busy, len(set(e[3] for e in inside))))
PY
cat gpurun_out/r02/overlap.txt
