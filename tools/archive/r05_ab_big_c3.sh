#!/bin/bash
# dev (GPU box): a whole configs[2] partition (and the MIXED one) under the plan's shape (2048 slots / 512 threads) and under the 4096-slot table with 512 threads
repo=${GRAFT_REPO_ROOT:-$PWD}
run() { lib=$1; env_shape=$2; shift 2
  KMD_TILE_SHAPE=$env_shape KMD_LIB=$repo/build_sweep/$lib.so python3 tools/kbench_pipeline.py --device --rows 39062500 --iters 6 "$@" 2>/dev/null | tail -1 | grep -o "kmd_merge_filter) [0-9.]* ms"; }
for rep in 1 2 3; do
  echo "C3    plan(small): $(run r5_base '')   big512 forced: $(run r5_big512 512x4096)   base big(1024) forced: $(run r5_base 1024x4096)"
  echo "MIXED plan(small): $(run r5_base '' --profile 1)   big512 forced: $(run r5_big512 512x4096 --profile 1)"
done
