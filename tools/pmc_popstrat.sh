#!/bin/bash
# dev tool (GPU box): FP64 instruction counts of the K3 pop-strat kernel (per launch, whole grid).
# usage: bash tools/pmc_popstrat.sh [kbench_popstrat args]
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf $repo/gpurun_out/pmc_popstrat_$tag
  (cd $repo && timeout 200 rocprofv3 --pmc $set -d gpurun_out/pmc_popstrat_$tag -o pmc --output-format csv -- python3 tools/kbench_popstrat.py "$@" > gpurun_out/pmc_popstrat.log 2>&1 < /dev/null)
  python3 - "$repo/gpurun_out/pmc_popstrat_$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        if 'popstrat' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    for c, v in d.items(): print(k, c, '%.5g per launch (%d launches)' % (v / n[(k, c)], n[(k, c)]))
PY
done
tail -5 $repo/gpurun_out/pmc_popstrat.log
