#!/bin/bash
# dev tool (GPU box): SQ counters of the wide merge kernel for library variants, one block per variant.
# usage: bash tools/pmc_ab.sh [-a "<kbench_pipeline args>"] [-k <kernel name substring>] build_sweep/a.so ...
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo"
extra=""
if [ "$1" = "-a" ]; then extra=$2; shift 2; fi
kern=""
if [ "$1" = "-k" ]; then kern=$2; shift 2; fi
for lib in "$@"; do
  tag=$(basename $lib .so)
  i=0
  for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
             "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_WAVES"; do
    i=$((i+1))
    rm -rf gpurun_out/pab_${tag}_$i
    KMD_LIB=$repo/$lib timeout 200 rocprofv3 --pmc $set -d gpurun_out/pab_${tag}_$i -o pmc --output-format csv -- python3 tools/kbench_pipeline.py --fused-only --iters 1 $extra > gpurun_out/pab_$tag.log 2>&1 < /dev/null
  done
  python3 - "$tag" "$kern" <<'PY'
import csv, glob, sys, collections
tag, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('gpurun_out/pab_%s_*/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if kern:
            if kern not in r['Kernel_Name']: continue
        elif 'k_tile_sums<512, 2048u, true, false, true' not in r['Kernel_Name']: continue      # (not the other shape's launch that only leaves again)
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
v = {c: acc[c] / n[c] for c in acc}
print('== %s %s' % (tag, kern))
print('  ' + '  '.join('%s %.4g' % (c, x) for c, x in sorted(v.items())))
cu = v.get('SQ_BUSY_CU_CYCLES', 0)
if cu:
    print('  per busy CU cycle: LDS %.0f%% (conflicts %.0f%% of it), SALU insts %.0f%%, VALU insts %.0f%%; wave cycles: waiting %.0f%%, issue-stalled %.0f%% (LDS %.0f%%)'
          % (100 * v.get('SQ_LDS_IDX_ACTIVE', 0) / cu, 100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, v.get('SQ_LDS_IDX_ACTIVE', 1)),
             100 * v.get('SQ_INSTS_SALU', 0) / cu, 100 * v.get('SQ_INSTS_VALU', 0) / cu,
             100 * v.get('SQ_WAIT_ANY', 0) / max(1, v.get('SQ_WAVE_CYCLES', 1)), 100 * v.get('SQ_WAIT_INST_ANY', 0) / max(1, v.get('SQ_WAVE_CYCLES', 1)),
             100 * v.get('SQ_WAIT_INST_LDS', 0) / max(1, v.get('SQ_WAVE_CYCLES', 1))))
PY
done
