#!/bin/bash
# dev (GPU box): the two table shapes against the partition's size (20v20 synthetic partitions built on the device; whole calls)
repo=${GRAFT_REPO_ROOT:-$PWD}
run() { env_shape=$1; shift
  KMD_TILE_SHAPE=$env_shape python3 tools/kbench_pipeline.py --device --iters 6 "$@" 2>/dev/null | tail -1 | grep -o "kmd_merge_filter) [0-9.]* ms"; }
for rows in 2000000 4000000 8000000 16000000 39062500; do
  for rep in 1 2; do
    echo "rows $rows: small $(run 512x2048 --rows $rows)   big $(run 1024x4096 --rows $rows)   plan $(run '' --rows $rows)"
  done
done
for rows in 8000000 39062500; do
  echo "MIXED rows $rows: small $(run 512x2048 --rows $rows --profile 1)   big $(run 1024x4096 --rows $rows --profile 1)"
  echo "100v100 rows $((rows / 5)): small $(run 512x2048 --rows $((rows / 5)) --nc 100 --nk 100)   big $(run 1024x4096 --rows $((rows / 5)) --nc 100 --nk 100)"
  echo "4v4 rows $((rows * 2)): small $(run 512x2048 --rows $((rows * 2)) --nc 4 --nk 4)   big $(run 1024x4096 --rows $((rows * 2)) --nc 4 --nk 4)"
done
