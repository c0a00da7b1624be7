rm -f gpurun_out/r05_ab/summary.txt
tools/archive/r05_ab.sh build_sweep/r5_base.so build_sweep/r5_jobptr.so
for pct in 50 56 62; do
  for a in "--sparse 0.1 --rows 40000000" "--sparse 0.3 --rows 13333333"; do
    echo "load $pct% $a: $(KMD_TILE_LOAD_PCT=$pct python3 tools/kbench_pipeline.py --fused-only $a --iters 4 2>/dev/null | tail -1 | cut -c1-150)"
  done
done
