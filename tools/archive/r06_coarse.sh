for c in 4096 1024 256; do
  echo "== KMD_TILE_COARSE=$c"
  KMD_TILE_COARSE=$c python3 tools/kbench_pipeline.py --fused-only --iters 8 2>&1 | grep -o "kmd_merge_filter) [0-9.]* ms"
  KMD_TILE_COARSE=$c python3 tools/kbench_pipeline.py --fused-only --sparse 0.1 --rows 40000000 --iters 4 2>&1 | grep -o "kmd_merge_filter) [0-9.]* ms"
  KMD_TILE_COARSE=$c python3 tools/kbench_pipeline.py --device --rows 39062500 --iters 5 2>&1 | grep -o "kmd_merge_filter) [0-9.]* ms"
done
