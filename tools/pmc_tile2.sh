#!/bin/bash
# dev tool (GPU box): utilisation counters of the fused merge kernel, normalised by busy CU cycles
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_NOT_TAKEN"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf $repo/gpurun_out/pmc2_$tag
  (cd $repo && timeout 200 rocprofv3 --pmc $set -d gpurun_out/pmc2_$tag -o pmc --output-format csv -- python3 tools/kbench_pipeline.py --fused-only --iters 1 "$@" > gpurun_out/pmc2.log 2>&1 < /dev/null)
  python3 - "$repo/gpurun_out/pmc2_$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_tile_sums' not in r['Kernel_Name'] or 'true, false, true' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for c, v in acc.items(): print(c, '%.4g per launch' % (v / n[c]))
PY
done
