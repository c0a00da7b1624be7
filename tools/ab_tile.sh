#!/bin/bash
# dev tool (GPU box): A/B of libkmdiff_hip.so against variant builds in build_sweep/ on the fused merge + test
# usage: bash tools/ab_tile.sh [-q] lib...      (-q: the 20v20 and the sparse case only)
cd $GRAFT_REPO_ROOT
cases=("" "--keys clustered" "--sparse 0.3 --rows 13333333" "--sparse 0.1 --rows 40000000" "--nc 4 --nk 4 --rows 20000000" "--nc 100 --nk 100 --rows 800000")
if [ "$1" = "-q" ]; then shift; cases=("" "--sparse 0.1 --rows 40000000" "--nc 100 --nk 100 --rows 800000"); fi
run() { echo "== $* $ARGS"; env "$@" timeout 300 python tools/kbench_pipeline.py --fused-only --iters 6 $ARGS 2>&1 | grep -E "fused|Error|error" | tail -1 | sed 's/.*kmd_merge_filter) //'; }
for v in "$@"; do
  for a in "${cases[@]}"; do ARGS="$a" run KMD_LIB=$v; done
done
