#!/bin/bash
# On the GPU box: every kernel micro-benchmark of profiles/r01_kbench.txt in one go.
#   gpurun -- 'bash tools/kbench_all.sh > gpurun_out/r01_kbench.txt 2>gpurun_out/kbench.err'
run() { timeout 200 python3 "$@" 2>/dev/null < /dev/null | tail -1; }
kb() { tag=$1; shift; run tools/kbench.py --iters 20 --tag "$tag" "$@"; }
kb tiled_20v20   --layout tiled
kb tiled_4v4     --layout tiled --nc 4 --nk 4 --rows 100000000
kb tiled_50v50   --layout tiled --nc 50 --nk 50 --rows 16000000
kb tiled_100v100 --layout tiled --nc 100 --nk 100 --rows 8000000
kb tiled_u16     --layout tiled --count-bytes 2
kb tiled_u8      --layout tiled --count-bytes 1
kb soa_20v20     --layout soa
kb rows_20v20    --layout rows
kb rows_4v4      --layout rows --nc 4 --nk 4 --rows 100000000
kb rows_50v50    --layout rows --nc 50 --nk 50 --rows 16000000
kb rows_100v100  --layout rows --nc 100 --nk 100 --rows 8000000
kb rows_u16      --layout rows --count-bytes 2
kb rows_21v21    --layout rows --nc 21 --nk 21
kb rows_3v3      --layout rows --nc 3 --nk 3 --rows 100000000
kb rows_u8       --layout rows --count-bytes 1
for k in even random clustered; do
  for s in 4 20 100; do
    KMD_MERGE_PATH=fast-only run tools/kbench_merge.py --nc $s --nk $s --rows $((80000000 / s)) --iters 3 --keys $k
  done
done
echo "$(KMD_MERGE_PATH=sort run tools/kbench_merge.py --iters 2 --keys random)   [KMD_MERGE_PATH=sort]"
KMD_MERGE_PATH=fast-only run tools/kbench_merge.py --limbs 2 --nc 20 --nk 20 --rows 4000000 --iters 3 --keys random
KMD_MERGE_PATH=fast-only run tools/kbench_merge.py --limbs 2 --nc 50 --nk 50 --rows 1600000 --iters 3 --keys random
echo "$(KMD_MERGE_PATH=sort run tools/kbench_merge.py --limbs 2 --nc 50 --nk 50 --rows 1600000 --iters 2 --keys random)   [KMD_MERGE_PATH=sort]"
run tools/kbench_popstrat.py
run tools/kbench_popstrat.py --thr 0.05
run tools/kbench_popstrat.py --nc 20 --nk 20 2>/dev/null
run tools/kbench_pca.py
run tools/kbench_pca.py --nc 100 --nk 100 --rows 8000000 --rate 0.01
# streams -> survivors: the matrix path (K2 + K1) and the sums path (K2s + K1s), two lines per run
run2() { timeout 300 python3 "$@" 2>/dev/null < /dev/null | tail -2; }
for k in even random clustered; do run2 tools/kbench_pipeline.py --keys $k; done
run2 tools/kbench_pipeline.py --nc 4 --nk 4 --rows 20000000
run2 tools/kbench_pipeline.py --nc 100 --nk 100 --rows 800000
run2 tools/kbench_pipeline.py --sparse 0.3 --rows 13333333 --iters 3
run2 tools/kbench_pipeline.py --sparse 0.1 --rows 40000000 --iters 3
KMD_MERGE_PATH=fast-only run tools/kbench_merge.py --sparse 0.1 --rows 40000000 --iters 3 --keys random
