#!/bin/bash
# dev (GPU box): the merge's run-time knobs on a whole configs[2] partition under the 4096-slot table (whole single calls, best of 6)
run() { env "$@" python3 tools/kbench_pipeline.py --device --rows 39062500 --iters 6 2>/dev/null | tail -1 | grep -o "kmd_merge_filter) [0-9.]* ms"; }
for rep in 1 2; do
  echo "default: $(run A=1)   load 54: $(run KMD_TILE_LOAD_PCT=54)   load 58: $(run KMD_TILE_LOAD_PCT=58)   load 46: $(run KMD_TILE_LOAD_PCT=46)"
  echo "xcd off: $(run KMD_TILE_XCD=0)   coarse 8192: $(run KMD_TILE_COARSE=8192 KMD_TILE_COARSE_CELLS=19)   coarse 2048: $(run KMD_TILE_COARSE=2048)"
done
