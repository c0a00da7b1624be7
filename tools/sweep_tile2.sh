#!/bin/bash
# dev tool (GPU box): environment sweeps of the fused merge + test
cd $GRAFT_REPO_ROOT
run() { echo "== $* $ARGS"; env "$@" timeout 300 python tools/kbench_pipeline.py --fused-only --iters 6 $ARGS 2>&1 | grep -E "fused|Error|error" | tail -1 | sed 's/.*kmd_merge_filter) //'; }
for a in "" "--overlap 6" "--sparse 0.1 --rows 40000000" "--nc 4 --nk 4 --rows 20000000" "--nc 100 --nk 100 --rows 800000"; do
  ARGS="$a"
  run KMD_TILE_LOAD_PCT=50
  run KMD_TILE_LOAD_PCT=40
  run KMD_TILE_LOAD_PCT=33
  run KMD_TILE_SHAPE=1024x4096 KMD_TILE_LOAD_PCT=50
  run KMD_TILE_SHAPE=1024x4096 KMD_TILE_LOAD_PCT=40
  run KMD_TILE_SHAPE=1024x4096 KMD_TILE_LOAD_PCT=33
done
