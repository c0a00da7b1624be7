#!/bin/bash
# one lease: tile-merge + threshold + bench + popstrat tests on the current tree, A/B of the sparse shapes, tile phases, the command end to end
python -m pytest tests/test_gpu_tilemerge.py tests/test_gpu_threshold.py tests/test_gpu_bench.py tests/test_gpu_pack.py -x -q -m gpu 2>&1 | tail -6
bash tools/r06_ab.sh -s "1 2 6" build_sweep/r6_base2.so build_sweep/r6_walk2.so build_sweep/r6_walk3.so
KMD_LIB=$PWD/build_sweep/r6_walk3_t.so timeout 300 python3 tools/kbench_pipeline.py --fused-only --sparse 0.1 --rows 40000000 --iters 1 2>&1 | grep "tile phases" | tail -6
python3 tools/cli_throughput.py --parts 8 --rows 2000000 --cpu-baseline --only "-t 16" 2>&1 | tail -12
