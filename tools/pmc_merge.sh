#!/bin/bash
# dev tool (GPU box): instruction mix of the K2 merge kernel.  usage: bash tools/pmc_merge.sh [kbench_merge args]
repo=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf $repo/gpurun_out/pmc_merge_$tag
  (cd $repo && KMD_MERGE_PATH=fast timeout 200 rocprofv3 --pmc $set -d gpurun_out/pmc_merge_$tag -o pmc --output-format csv -- python3 tools/kbench_merge.py --iters 1 "$@" > gpurun_out/pmc_merge.log 2>&1 < /dev/null)
  python3 - "$repo/gpurun_out/pmc_merge_$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        if 'bucket_merge' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    for c, v in d.items(): print(c, '%.4g per launch' % (v / n[(k, c)]))
PY
done
