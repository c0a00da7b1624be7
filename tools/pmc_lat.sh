#!/bin/bash
# dev tool (GPU box): average LDS / vector-memory instruction latency of the wide merge kernel (level counters / instructions)
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo"
for lib in "$@"; do
  tag=$(basename $lib .so)
  i=0
  for set in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD" "SQ_LEVEL_WAVES SQ_LDS_ATOMIC_RETURN SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_STORE"; do
    i=$((i+1))
    rm -rf gpurun_out/plat_${tag}_$i
    KMD_LIB=$repo/$lib timeout 200 rocprofv3 --pmc $set -d gpurun_out/plat_${tag}_$i -o pmc --output-format csv -- python3 tools/kbench_pipeline.py --fused-only --iters 1 > gpurun_out/plat_$tag.log 2>&1 < /dev/null
  done
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('gpurun_out/plat_%s_*/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_tile_sums' not in r['Kernel_Name'] or 'true, false, true' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
v = {c: acc[c] / n[c] for c in acc}
print('== %s  ' % tag + '  '.join('%s %.4g' % (c, x) for c, x in sorted(v.items())))
if v.get('SQ_INSTS_LDS'): print('   LDS level / insts = %.1f ; VMEM level / insts = %.1f' % (v.get('SQ_INST_LEVEL_LDS', 0) / v['SQ_INSTS_LDS'], v.get('SQ_INST_LEVEL_VMEM', 0) / max(1, v.get('SQ_INSTS_VMEM', 1))))
PY
done
