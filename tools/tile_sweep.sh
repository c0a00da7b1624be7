#!/bin/bash
# A/B of the fused merge + test kernel (kmd_merge_filter) on one partition; KMD_DEBUG prints the plan
cd "$(dirname "$0")/.."
run() { echo "== $* $ARGS"; env "$@" timeout 300 python tools/kbench_pipeline.py --fused-only --iters 4 $ARGS 2>&1 | grep -E "fused|Error|error|level 0" | tail -2; }
ARGS="" run KMD_DEBUG=1
ARGS="" run KMD_TILE_XCD=0
ARGS="" run KMD_TILE_LOAD_PCT=25
ARGS="" run KMD_TILE_LOAD_PCT=45
ARGS="" run KMD_TILE_SHAPE=1024x4096
ARGS="" run KMD_TILE_SHAPE=1024x2048
ARGS="--sparse 0.3" run KMD_DEBUG=1
ARGS="--sparse 0.1" run KMD_DEBUG=1
ARGS="--sparse 0.1" run KMD_TILE_SHAPE=1024x4096
ARGS="--nc 4 --nk 4" run KMD_DEBUG=1
ARGS="--nc 100 --nk 100 --rows 1000000" run KMD_DEBUG=1
ARGS="--keys clustered" run KMD_DEBUG=1
