for c in 512 1024 2048 4096; do for cells in 16 17 18; do
  echo "== coarse $c cells 2^$cells"
  for a in "" "--sparse 0.1 --rows 40000000 --iters 3" "--nc 100 --nk 100 --rows 800000"; do
    KMD_TILE_COARSE=$c KMD_TILE_COARSE_CELLS=$cells bash tools/ab_tile3.sh -a "$a" kmdiff_amd/lib/libkmdiff_hip.so | grep -o "kmd_merge_filter) [0-9.]* ms\|k_tile_probe [0-9]*x [0-9.]*us\|k_tile_bounds [0-9]*x [0-9.]*us" | tr '\n' ' '; echo
  done
done; done
