for pct in 50 60 70 80; do
  echo "== load $pct"
  KMD_TILE_LOAD_PCT=$pct python3 tools/kbench_pipeline.py --fused-only --sparse 0.1 --rows 40000000 --iters 4 2>/dev/null | tail -1 | cut -c1-200
  KMD_TILE_LOAD_PCT=$pct python3 tools/kbench_pipeline.py --fused-only --sparse 0.3 --rows 13333333 --iters 4 2>/dev/null | tail -1 | cut -c1-200
  KMD_TILE_LOAD_PCT=$pct python3 tools/kbench_pipeline.py --device --rows 39062500 --iters 4 2>/dev/null | tail -1 | cut -c1-200
  KMD_TILE_LOAD_PCT=$pct python3 tools/kbench_pipeline.py --device --rows 100000000 --nc 4 --nk 4 --iters 4 2>/dev/null | tail -1 | cut -c1-200
done
