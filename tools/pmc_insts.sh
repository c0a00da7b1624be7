#!/bin/bash
# dev tool (GPU box): instruction counts of the merge kernel for a given library (KMD_LIB)
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $repo/gpurun_out/pmc_i
(cd $repo && timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH -d gpurun_out/pmc_i -o pmc --output-format csv -- python3 tools/kbench_pipeline.py --fused-only --iters 1 "$@" > gpurun_out/pmc_i.log 2>&1 < /dev/null)
python3 - "$repo/gpurun_out/pmc_i" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_tile_sums' not in r['Kernel_Name'] or 'true, false, true' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print('  '.join('%s %.4g' % (c, v / n[c]) for c, v in sorted(acc.items())))
PY
