# dev tool (GPU box): the two table shapes of the fused merge on partitions of different records-per-row
for a in "" "--sparse 0.6 --rows 6666666" "--sparse 0.3 --rows 13333333" "--sparse 0.2 --rows 20000000" "--sparse 0.1 --rows 40000000" "--nc 4 --nk 4 --rows 20000000" "--nc 100 --nk 100 --rows 800000"; do
 echo "== $a"
 for sh in 512x2048 1024x4096; do
  KMD_TILE_SHAPE=$sh bash tools/ab_tile3.sh -a "$a --iters 3" kmdiff_amd/lib/libkmdiff_hip.so | sed "s/^libkmdiff_hip/$sh/" | cut -c1-150
 done
done
