#!/bin/bash
# dev tool (GPU box): A/B of libkmdiff_hip.so against variant builds in build_sweep/ on the fused merge + test
cd $GRAFT_REPO_ROOT
run() { echo "== $* $ARGS"; env "$@" timeout 300 python tools/kbench_pipeline.py --fused-only --iters 6 $ARGS 2>&1 | grep -E "fused|Error|error" | tail -1 | sed 's/.*kmd_merge_filter) //'; }
for v in "$@"; do
  for a in "" "--keys clustered" "--sparse 0.3 --rows 13333333" "--sparse 0.1 --rows 40000000" "--nc 4 --nk 4 --rows 20000000" "--nc 100 --nk 100 --rows 800000"; do
    ARGS="$a" run KMD_LIB=$v
  done
done
