for shape in 1024x4096 512x2048; do for pct in 25 33 40 50; do
  export KMD_TILE_SHAPE=$shape KMD_TILE_LOAD_PCT=$pct
  echo "== $shape load $pct"; bash tools/ab_tile3.sh build_sweep/a1.so 2>&1 | cut -c1-125
done; done
