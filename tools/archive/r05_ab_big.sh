#!/bin/bash
# dev (GPU box): the 4096-slot table under 512 threads (eight waves of 128 registers) against 1024 (sixteen of 64)
repo=${GRAFT_REPO_ROOT:-$PWD}
KMD_LIB=$repo/build_sweep/r5_big512.so python -m pytest tests/test_gpu_tilemerge.py -x -q -m gpu 2>&1 | tail -1
run() { # lib shape-env args...
  lib=$1; env_shape=$2; shift 2
  KMD_TILE_SHAPE=$env_shape KMD_LIB=$repo/build_sweep/$lib.so python3 tools/kbench_pipeline.py --fused-only "$@" 2>/dev/null | tail -1 | grep -o "kmd_merge_filter) [0-9.]* ms"
}
for rep in 1 2; do
for v in base:1024x4096 big512:512x4096 big512r8:512x4096; do
  lib=r5_${v%%:*}; shp=${v#*:}
  echo "$lib  2.9/row plan: $(run $lib '' --sparse 0.1 --rows 40000000 --iters 4)  7.8/row plan: $(run $lib '' --sparse 0.3 --rows 13333333 --iters 4)  7.8/row big: $(run $lib $shp --sparse 0.3 --rows 13333333 --iters 4)  4M big: $(run $lib $shp --iters 5)  4M plan: $(run $lib '' --iters 5)"
done
done
