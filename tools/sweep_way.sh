# dev tool (GPU box): many samples, rows of few records -- whole waves per run (KMD_TILE_G=6) at several table loads
# (slow: the partitions are made on the host, minutes each -- 40 GPU-minutes for the three below)
for a in "--nc 100 --nk 100 --rows 50000000 --sparse 0.015" "--nc 300 --nk 300 --rows 20000000 --sparse 0.012" "--nc 300 --nk 300 --rows 40000000 --sparse 0.006"; do
  echo "== $a"
  for lp in 50 45 40; do for g in 0 6; do
    KMD_TILE_LOAD_PCT=$lp KMD_TILE_G=$g KMD_DEBUG=1 timeout 900 python tools/kbench_pipeline.py --fused-only --iters 2 $a 2>&1 | grep -E "level 0:|pipeline" | tail -2 | sed "s/^/load $lp G=$g /" | cut -c1-60,120-200 | tr '\n' ' '; echo
  done; done
done
