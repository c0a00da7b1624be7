#!/bin/bash
# dev tool (GPU box): memory-side counters of one kernel (name substring) on the fused merge + test.
# usage: bash tools/pmc_mem.sh [-a "<kbench_pipeline args>"] <kernel substring> lib.so ...
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo"
extra=""
if [ "$1" = "-a" ]; then extra=$2; shift 2; fi
kern=$1; shift
for lib in "$@"; do
  tag=$(basename $lib .so)
  i=0
  for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    rm -rf gpurun_out/pmem_${tag}_$i
    KMD_LIB=$repo/$lib timeout 200 rocprofv3 --pmc $set -d gpurun_out/pmem_${tag}_$i -o pmc --output-format csv -- python3 tools/kbench_pipeline.py --fused-only --iters 1 $extra > gpurun_out/pmem_$tag.log 2>&1 < /dev/null
  done
  python3 - "$tag" "$kern" <<'PY'
import csv, glob, sys, collections
tag, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('gpurun_out/pmem_%s_*/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if kern not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print('== %s %s' % (tag, kern))
print('  ' + '  '.join('%s %.4g' % (c, acc[c] / n[c]) for c in sorted(acc)))
PY
done
