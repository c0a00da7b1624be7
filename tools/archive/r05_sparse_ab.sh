#!/bin/bash
# dev (GPU box): where the fused merge's time goes on rows of few records -- kernel trace of ablated builds, phase cycles of
# one tile (KMD_TILE_TIMING), SQ counters of the default build.   usage: tools/archive/r05_sparse_ab.sh [variants...]
repo=${GRAFT_REPO_ROOT:-$PWD}
out=gpurun_out/sparse_ab; mkdir -p $out
A1="--sparse 0.1 --rows 40000000 --iters 3"
A2="--sparse 0.3 --rows 13333333 --iters 3"
vars=${@:-r5_base r5_ab16 r5_ab4 r5_ab20 r5_ab128}
for shape in 1 2; do
  eval args=\$A$shape
  echo "== shape $shape: $args" | tee -a $out/summary.txt
  bash tools/ab_tile3.sh -a "$args" $(for v in $vars; do echo build_sweep/$v.so; done) 2>&1 | tee -a $out/summary.txt
done
if [ -f build_sweep/r5_timing.so ]; then
  for shape in 1 2; do
    eval args=\$A$shape
    KMD_LIB=$repo/build_sweep/r5_timing.so timeout 300 python3 tools/kbench_pipeline.py --fused-only $args > $out/timing_$shape.log 2>&1
    grep -h "tile phases\|tile timing\|walk barrier" $out/timing_$shape.log | sort | uniq -c | sort -rn | head -12 | tee -a $out/summary.txt
  done
fi
bash tools/pmc_ab.sh -a "$A1" -k "k_tile_sums<1024" build_sweep/r5_base.so 2>&1 | tee -a $out/summary.txt
bash tools/pmc_ab.sh -a "$A2" -k "k_tile_sums<" build_sweep/r5_base.so 2>&1 | tee -a $out/summary.txt
