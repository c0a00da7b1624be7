#!/bin/bash
# On the GPU box (gpurun -- 'bash tools/refresh_profiles_r03.sh'): the evidence profiles/r03_* is built from.
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo" && mkdir -p gpurun_out/r03 && O=gpurun_out/r03
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1 < /dev/null
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_r03.json 2> $O/bench_r03.err < /dev/null
rm -rf $O/prof_bench $O/prof_pipe $O/prof_sparse $O/prof_batch
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline \
  > $O/prof_bench.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pipe -o pipe -- python3 tools/kbench_pipeline.py --fused-only --iters 8 \
  > $O/prof_pipe.log 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sparse -o sparse -- python3 tools/kbench_pipeline.py --fused-only --iters 4 --sparse 0.1 --rows 40000000 \
  > $O/prof_sparse.log 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_batch -o batch -- python3 tools/kbench_batch.py --iters 3 \
  > $O/prof_batch.log 2>&1 < /dev/null
# HBM traffic of the merge kernel: FETCH_SIZE / WRITE_SIZE in passes of their own
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 tools/kbench_pipeline.py --fused-only --iters 2 > $O/pmc_$c.log 2>&1 < /dev/null
done
bash tools/pmc_ab.sh kmdiff_amd/lib/libkmdiff_hip.so > $O/pmc_tile.txt 2>&1
bash tools/pmc_popstrat.sh --thr 0.05 > $O/pmc_popstrat.txt 2>&1
{
  run() { timeout 400 python3 "$@" 2>/dev/null < /dev/null | grep -E "fused|popstrat|merge\+filter|batch" | tail -2; }
  for k in random even clustered; do run tools/kbench_pipeline.py --fused-only --keys $k; done
  run tools/kbench_pipeline.py --fused-only --nc 4 --nk 4 --rows 20000000
  run tools/kbench_pipeline.py --fused-only --nc 50 --nk 50 --rows 1600000
  run tools/kbench_pipeline.py --fused-only --nc 100 --nk 100 --rows 800000
  run tools/kbench_pipeline.py --fused-only --sparse 0.6 --rows 6666666 --iters 3
  run tools/kbench_pipeline.py --fused-only --sparse 0.3 --rows 13333333 --iters 3
  run tools/kbench_pipeline.py --fused-only --sparse 0.2 --rows 20000000 --iters 3
  run tools/kbench_pipeline.py --fused-only --sparse 0.1 --rows 40000000 --iters 3
  run tools/kbench_pipeline.py --fused-only --limbs 2
  run tools/kbench_pipeline.py --fused-only --limbs 2 --nc 50 --nk 50 --rows 1600000
  run tools/kbench_pipeline.py --fused-only --overlap 6
  run tools/kbench_batch.py --parts 6
  run tools/kbench_batch.py --parts 12
  run tools/kbench_batch.py --parts 24
  run tools/kbench_popstrat.py
  run tools/kbench_popstrat.py --thr 0.05
  run tools/kbench_popstrat.py --nc 20 --nk 20
} > $O/kbench.txt 2>&1
cp gpurun_out/r03/kbench_k1.txt $O/kbench_k1.txt 2>/dev/null
{
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
  kb rows_21v21    --layout rows --nc 21 --nk 21
} > $O/kbench_k1.txt 2>&1
timeout 900 python3 tools/cli_throughput.py --rows 2000000 --parts 8 > $O/cli_throughput.txt 2>&1 < /dev/null
ls $O
