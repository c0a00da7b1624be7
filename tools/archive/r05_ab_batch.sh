#!/bin/bash
# dev (GPU box): partitions in flight inside kmd_merge_filter_batch (KMD_BATCH_STREAMS): whole configs[2] partitions, 16 M-row, 4 M-row, the MIXED one
repo=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for v in base bs2 bs3; do
  a=$(KMD_LIB=$repo/build_sweep/r5_$v.so python3 tools/kbench_batch.py --device --rows 39062500 --parts 12 2>/dev/null | tail -1 | grep -o "kmd_merge_filter_batch [0-9.]* ms")
  b=$(KMD_LIB=$repo/build_sweep/r5_$v.so python3 tools/kbench_batch.py --device --rows 16000000 --parts 12 2>/dev/null | tail -1 | grep -o "kmd_merge_filter_batch [0-9.]* ms")
  c=$(KMD_LIB=$repo/build_sweep/r5_$v.so python3 tools/kbench_batch.py --parts 12 2>/dev/null | tail -1 | grep -o "kmd_merge_filter_batch [0-9.]* ms")
  d=$(KMD_LIB=$repo/build_sweep/r5_$v.so python3 tools/kbench_batch.py --device --rows 8000000 --parts 12 2>/dev/null | tail -1 | grep -o "kmd_merge_filter_batch [0-9.]* ms")
  echo "$v: 39M $a | 16M $b | 8M $d | 4M $c"
done
done
