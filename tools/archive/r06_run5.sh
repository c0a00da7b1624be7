#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-$PWD}
rm -rf gpurun_out/k2trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k2trace -o t -- python3 tools/kbench_merge.py --keys random > gpurun_out/k2trace.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/k2trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print('%-90s %4s x %9.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
tail -1 gpurun_out/k2trace.log
python3 tools/kbench_merge.py --nc 100 --nk 100 --rows 1000000 | tail -1
python3 tools/kbench_merge.py --sparse 0.1 --rows 20000000 | tail -1
python3 tools/kbench_merge.py --limbs 2 | tail -1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge" 2>&1 | tail -3
