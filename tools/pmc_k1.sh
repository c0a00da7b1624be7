#!/bin/bash
# dev tool (GPU box): SQ counters of a K1 / K1r kernel.  usage: bash tools/pmc_k1.sh <kernel name substring> <kbench.py args...>
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo"
kern=$1; shift
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf gpurun_out/pk1_$i
  timeout 200 rocprofv3 --pmc $set -d gpurun_out/pk1_$i -o pmc --output-format csv -- python3 tools/kbench.py --iters 2 "$@" > gpurun_out/pk1.log 2>&1 < /dev/null
done
python3 - "$kern" <<'PY'
import csv, glob, sys, collections
kern = sys.argv[1]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('gpurun_out/pk1_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if kern not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
v = {c: acc[c] / n[c] for c in acc}
print('== %s' % kern)
print('  ' + '  '.join('%s %.4g' % (c, x) for c, x in sorted(v.items())))
cu = v.get('SQ_BUSY_CU_CYCLES', 0)
if cu:
    print('  per busy CU cycle: LDS %.0f%% (conflicts %.0f%% of it), SALU insts %.0f%%, VALU insts %.0f%%; wave cycles: waiting %.0f%%, issue-stalled %.0f%% (LDS %.0f%%)'
          % (100 * v.get('SQ_LDS_IDX_ACTIVE', 0) / cu, 100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, v.get('SQ_LDS_IDX_ACTIVE', 1)),
             100 * v.get('SQ_INSTS_SALU', 0) / cu, 100 * v.get('SQ_INSTS_VALU', 0) / cu,
             100 * v.get('SQ_WAIT_ANY', 0) / max(1, v.get('SQ_WAVE_CYCLES', 1)), 100 * v.get('SQ_WAIT_INST_ANY', 0) / max(1, v.get('SQ_WAVE_CYCLES', 1)),
             100 * v.get('SQ_WAIT_INST_LDS', 0) / max(1, v.get('SQ_WAVE_CYCLES', 1))))
PY
