#!/bin/bash
# dev (GPU box): an 8192-slot table (one 1024-thread workgroup per CU, 128 VGPRs) against the 4096-slot one (two per CU, 64 VGPRs)
repo=${GRAFT_REPO_ROOT:-$PWD}
run() { lib=$1; env_shape=$2; shift 2
  KMD_TILE_SHAPE=$env_shape KMD_LIB=$repo/build_sweep/$lib.so python3 tools/kbench_pipeline.py "$@" 2>/dev/null | tail -1 | grep -o "kmd_merge_filter) [0-9.]* ms.*sig=[0-9]*" | sed 's/  [0-9.e+]* rows.s.*sig/ sig/'; }
for rep in 1 2; do
  echo "2.9/row: base(plan=4096) $(run r5_base '' --fused-only --sparse 0.1 --rows 40000000 --iters 4) | 8192 $(run r5_huge '' --fused-only --sparse 0.1 --rows 40000000 --iters 4)"
  echo "7.8/row: base(plan=2048) $(run r5_base '' --fused-only --sparse 0.3 --rows 13333333 --iters 4) | base 4096 $(run r5_base 1024x4096 --fused-only --sparse 0.3 --rows 13333333 --iters 4) | 8192 $(run r5_huge 1024x8192 --fused-only --sparse 0.3 --rows 13333333 --iters 4)"
  echo "C3:      base(plan=4096) $(run r5_base '' --device --rows 39062500 --iters 5) | 8192 $(run r5_huge 1024x8192 --device --rows 39062500 --iters 5)"
  echo "MIXED:   base(plan=4096) $(run r5_base '' --device --rows 39062500 --iters 5 --profile 1) | 8192 $(run r5_huge 1024x8192 --device --rows 39062500 --iters 5 --profile 1)"
done
